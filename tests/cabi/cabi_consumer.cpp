// A consumer of include/samrs_hip.h that knows nothing of Python or torch: the drop-in boundary used the way a C / C++ host would
// (SURVEY.md 8b).  Built by tests/test_cabi_consumer_gpu.py with hipcc against the PUBLIC header only and linked with libsamrs_hip.so.
//
//   cabi_consumer <dir>    reads   <dir>/manifest.txt      one line per tensor: name ndim d0 d1 ... offset_in_floats
//                                  <dir>/weights.bin       fp32, concatenated
//                                  <dir>/config.txt        embed_dim depth num_heads n_global g0..g3 img patch window out_chans
//                                  <dir>/image.u8          1024 x 1024 x 3 uint8
//                                  <dir>/boxes.f32         n x 4 fp32 xyxy (input frame), <dir>/labels.i32  n x int32
//                          writes  <dir>/masks.u8 (n x 1024 x 1024), <dir>/iou.f32, <dir>/seg.u8 (the painted class map), <dir>/areas.i64
// Mirrors Generate Dataset/main_sam_hbox_semantic.py:87-89,155,174-206: build, set_image, predict on the boxes, paint in box order.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "samrs_hip.h"

#define HIPOK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); return 2; } } while (0)
#define OK(e, x) do { int r_ = (x); if (r_ != SAMRS_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, samrs_last_error(e)); return 3; } } while (0)

template <class T>
static std::vector<T> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(4); }
    const size_t bytes = (size_t)f.tellg();
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)bytes);
    return v;
}
template <class T>
static void dump(const std::string& path, const std::vector<T>& v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    const std::string dir = argv[1];
    if (samrs_abi_version() != SAMRS_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    samrs_config cfg{};
    {
        std::ifstream f(dir + "/config.txt");
        f >> cfg.embed_dim >> cfg.depth >> cfg.num_heads >> cfg.n_global;
        for (int i = 0; i < cfg.n_global; ++i) f >> cfg.global_attn_indexes[i];
        f >> cfg.img_size >> cfg.patch_size >> cfg.window_size >> cfg.out_chans;
    }
    cfg.max_images = 1; cfg.max_prompts = 8; cfg.max_points = 1; cfg.precision = SAMRS_PREC_F16;
    char err[512] = {0};
    samrs_engine_t* e = samrs_create(&cfg, 0, err, sizeof(err));
    if (!e) { fprintf(stderr, "samrs_create: %s\n", err); return 1; }
    const std::vector<float> weights = slurp<float>(dir + "/weights.bin");
    {
        std::ifstream f(dir + "/manifest.txt");
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ls(line);
            std::string name; int ndim; ls >> name >> ndim;
            std::vector<int64_t> shape(ndim);
            for (auto& d : shape) ls >> d;
            size_t off; ls >> off;
            OK(e, samrs_load_weight(e, name.c_str(), weights.data() + off, shape.data(), ndim));
        }
    }
    hipStream_t s;
    HIPOK(hipStreamCreate(&s));
    OK(e, samrs_finalize_weights(e, s));                                        // strict: fails on a missing / misshapen tensor
    const std::vector<uint8_t> image = slurp<uint8_t>(dir + "/image.u8");
    const std::vector<float> boxes = slurp<float>(dir + "/boxes.f32");
    const std::vector<int32_t> labels = slurp<int32_t>(dir + "/labels.i32");
    const int n = (int)labels.size(), H = cfg.img_size, W = cfg.img_size;
    uint8_t *d_img, *d_masks, *d_seg; float *d_boxes, *d_iou; int32_t* d_lab; int64_t* d_area;
    HIPOK(hipMalloc(&d_img, image.size())); HIPOK(hipMalloc(&d_masks, (size_t)n * H * W)); HIPOK(hipMalloc(&d_seg, (size_t)H * W));
    HIPOK(hipMalloc(&d_boxes, boxes.size() * 4)); HIPOK(hipMalloc(&d_iou, n * 4)); HIPOK(hipMalloc(&d_lab, n * 4)); HIPOK(hipMalloc(&d_area, n * 8));
    HIPOK(hipMemcpyAsync(d_img, image.data(), image.size(), hipMemcpyHostToDevice, s));
    HIPOK(hipMemcpyAsync(d_boxes, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice, s));
    HIPOK(hipMemcpyAsync(d_lab, labels.data(), n * 4, hipMemcpyHostToDevice, s));
    HIPOK(hipMemsetAsync(d_seg, 255, (size_t)H * W, s));                        // main_sam_hbox_semantic.py:162
    OK(e, samrs_set_images(e, d_img, 1, H, W, 0, s));                           // :155
    OK(e, samrs_predict(e, 0, n, d_boxes, nullptr, nullptr, 0, nullptr, 0, 0, H, W, H, W, d_masks, d_iou, nullptr, s));   // :176-181 (chunked inside: n > max_prompts)
    OK(e, samrs_paint(e, d_masks, d_lab, n, H, W, d_seg, d_area, nullptr, nullptr, 18, s));                                // :195-206
    std::vector<uint8_t> masks((size_t)n * H * W), seg((size_t)H * W);
    std::vector<float> iou(n);
    std::vector<int64_t> area(n);
    HIPOK(hipMemcpyAsync(masks.data(), d_masks, masks.size(), hipMemcpyDeviceToHost, s));
    HIPOK(hipMemcpyAsync(seg.data(), d_seg, seg.size(), hipMemcpyDeviceToHost, s));
    HIPOK(hipMemcpyAsync(iou.data(), d_iou, n * 4, hipMemcpyDeviceToHost, s));
    HIPOK(hipMemcpyAsync(area.data(), d_area, n * 8, hipMemcpyDeviceToHost, s));
    HIPOK(hipStreamSynchronize(s));                                             // the only synchronisation: the caller's (.cpu() in the reference, :189)
    dump(dir + "/masks.u8", masks); dump(dir + "/seg.u8", seg); dump(dir + "/iou.f32", iou); dump(dir + "/areas.i64", area);
    samrs_destroy(e);
    printf("cabi_consumer: %d boxes decoded and painted through include/samrs_hip.h (ABI %d)\n", n, samrs_abi_version());
    return 0;
}
