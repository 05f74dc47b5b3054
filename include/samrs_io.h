/* libsamrs_io.so -- host-side image IO of the generation driver (C ABI, no GPU, no torch types).
 *
 * What it replaces in the reference: the per-image file IO either side of the hot path in
 * `Generate Dataset/main_sam_hbox_semantic.py`:
 *   :114      img = np.array(Image.open(path))                                   -> samrs_io_png_read_rgb
 *   :212-215  Image.fromarray(seg_mask).save(gray/<stem>.png), likewise color/   -> samrs_io_png_write_gray / _write_lut_rgb
 * The reference does these serially on the one Python thread that also drives the GPU; here they run on reader / writer
 * threads, and because every entry point is a plain C call the Python caller (ctypes) holds no interpreter lock while
 * one runs -- a PIL encode re-takes it per 64 KiB block, which is what capped `python -m samrs_amd.generate` at about a third
 * of the pipeline's rate.
 *
 * PNG subset: 8 bits per sample, non-interlaced.  Read: colour types 0 (gray), 2 (RGB), 3 (palette), 4 (gray + alpha),
 * 6 (RGBA); alpha / tRNS are dropped (what PIL's `.convert("RGB")` does).  Anything else returns SAMRS_IO_UNSUPPORTED and the
 * caller decodes with PIL.  Written files are plain PNGs (one IDAT, adaptive row filters, zlib level `level`): any reader
 * returns the same pixels the reference's files hold; the compressed bytes are not part of the contract.
 *
 * Every function returns 0 on success or a negative SAMRS_IO_* code; none of them throws or aborts.
 */
#ifndef SAMRS_IO_H
#define SAMRS_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMRS_IO_ABI_VERSION 3

#define SAMRS_IO_OK            0
#define SAMRS_IO_EOPEN        -1   /* cannot open / create the file (errno is left set) */
#define SAMRS_IO_UNSUPPORTED  -2   /* a valid file this subset does not cover (not PNG, 16-bit, interlaced, < 8 bit) */
#define SAMRS_IO_ECORRUPT     -3   /* truncated file, bad CRC, bad zlib stream, wrong amount of pixel data */
#define SAMRS_IO_ESIZE        -4   /* destination too small / zero or oversized dimensions */
#define SAMRS_IO_EWRITE       -5   /* short write, or the rename onto the final path failed */
#define SAMRS_IO_ENOMEM       -6

/* `level` of the writers: 1..9 = zlib level with the default strategy (anything else is read as 6), or
 * SAMRS_IO_LEVEL_RUNS = zlib's run-length strategy: for class maps (long runs of one id; after row filtering, runs of zero) it is
 * about 2.5x faster than level 6 and no larger.  The pixels a reader gets back never depend on it. */
#define SAMRS_IO_LEVEL_RUNS   -1
/* SAMRS_IO_LEVEL_LABELS (class maps and their palette images only: samrs_io_png_write_gray / _write_lut_rgb): the label-aware
 * encoder.  The LZ77 parse is done on the 1-byte label map -- a run equal to the row above, a run equal to the left neighbour, or a
 * literal pixel -- and written as deflate tokens of the pixel stream (filter 0, dynamic Huffman): 5 - 10x less CPU than zlib level 6 on
 * the truecolour image, whose bytes zlib would have to match one by one.  Any PNG reader decodes the same pixels.
 * samrs_io_png_write_label_pair writes both images of a tile from one parse. */
#define SAMRS_IO_LEVEL_LABELS -2

int samrs_io_abi_version(void);

/* Width / height of a PNG without decoding it. */
int samrs_io_png_info(const char* path, int* height, int* width);

/* Decode `path` into `dst` as packed RGB (row stride 3 * width bytes).  `dst_bytes` must be >= 3 * h * w. */
int samrs_io_png_read_rgb(const char* path, uint8_t* dst, size_t dst_bytes, int* height, int* width);

/* The same from a memory buffer holding the whole file. */
int samrs_io_png_decode_rgb(const uint8_t* file, size_t file_bytes, uint8_t* dst, size_t dst_bytes, int* height, int* width);

/* Write an 8-bit gray PNG (the class map, 255 = unlabeled).  `stride` = bytes between rows of `src` (>= width).
 * The file is written to `<path>.tmp.<pid>` and renamed, so a reader never sees half a file and two processes never share a tmp. */
int samrs_io_png_write_gray(const char* path, const uint8_t* src, int height, int width, size_t stride, int level);

/* Write a truecolour PNG whose pixel is lut[3 * src[y, x] .. +3]: the palette image of the class map without
 * materialising the RGB array (main_sam_hbox_semantic.py:163,199 paint it on the host). `lut` has 256 * 3 bytes. */
int samrs_io_png_write_lut_rgb(const char* path, const uint8_t* src, int height, int width, size_t stride,
                               const uint8_t* lut, int level);

/* Both images of a class map from ONE parse (SAMRS_IO_LEVEL_LABELS encoder): gray_path = the 8-bit gray PNG, color_path = the palette
 * image lut[src[y, x]] -- main_sam_hbox_semantic.py:212-215 saves exactly this pair per tile.  The label runs are found once (tuned for
 * the 3-byte stream; the gray stream writes runs shorter than deflate's 3-byte minimum as literals) and written twice with each stream's
 * own Huffman codes.  Same pixels as the two single calls; each file atomic on its own (gray first). */
int samrs_io_png_write_label_pair(const char* gray_path, const char* color_path, const uint8_t* src, int height, int width, size_t stride,
                                  const uint8_t* lut);

/* Write a packed RGB array (stride in bytes). */
int samrs_io_png_write_rgb(const char* path, const uint8_t* src, int height, int width, size_t stride, int level);

#ifdef __cplusplus
}
#endif
#endif
