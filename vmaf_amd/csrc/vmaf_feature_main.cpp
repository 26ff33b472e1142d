// vmaf_feature_main.cpp - the reference's per-feature tool `feature/vmaf app fmt ref dis w h` (feature/src/vmaf_main.c:34-180)
// on the MI355X engine: same argv, same stdout grammar ("<atom>: <frame> <%f>" lines, feature/src/all.c:138-226,
// adm.c:351-361, vif.c:373-382, motion.c:182, ansnr.c:204-206), so that the legacy Python feature extractor
// (python/src/vmaf/core/feature_extractor.py: it parses these lines) can sit on top.  Frames are read raw and widened
// on the GPU; apps: adm, ansnr, motion, vif, all.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vmaf_hip.h"

namespace {

void usage()
{
    std::puts("usage: vmaf app fmt ref dis w h\n"
              "apps:\n\tadm\n\tansnr\n\tmotion\n\tvif\n\tall\n"
              "fmts:\n\tyuv420p\n\tyuv422p\n\tyuv444p\n\tyuv420p10le\n\tyuv422p10le\n\tyuv444p10le");
}

struct Plane { FILE *f = nullptr; size_t luma_bytes = 0, chroma_bytes = 0; };

// frame.c:44-124: luma of one frame, then skip the chroma; 0 ok, 2 end of stream, 1 error
int read_luma(Plane &p, void *dst)
{
    if (std::fread(dst, 1, p.luma_bytes, p.f) != p.luma_bytes) return std::feof(p.f) ? 2 : 1;
    if (std::fseek(p.f, (long)p.chroma_bytes, SEEK_CUR) != 0) {
        std::vector<char> tmp(1 << 20);
        for (size_t left = p.chroma_bytes; left;) {
            const size_t n = left < tmp.size() ? left : tmp.size();
            if (std::fread(tmp.data(), 1, n, p.f) != n) break;
            left -= n;
        }
    }
    return 0;
}

}  // namespace

int main(int argc, char *argv[])
{
    if (argc < 7) { usage(); return 2; }
    const std::string app = argv[1], fmt = argv[2];
    const char *ref_path = argv[3], *dis_path = argv[4];
    const int w = std::atoi(argv[5]), h = std::atoi(argv[6]);
    if (w <= 0 || h <= 0) { usage(); return 2; }
    const bool do_adm = app == "adm" || app == "all", do_vif = app == "vif" || app == "all";
    const bool do_motion = app == "motion" || app == "all", do_ansnr = app == "ansnr" || app == "all";
    if (!(do_adm || do_vif || do_motion || do_ansnr)) { usage(); return 2; }

    const size_t n = (size_t)w * h;
    size_t chroma;
    const std::string base = fmt.size() > 4 && fmt.compare(fmt.size() - 4, 4, "10le") == 0 ? fmt.substr(0, fmt.size() - 4) : fmt;
    const int bps = base == fmt ? 1 : 2;
    if (base == "yuv420p") {
        if (n % 2) { std::fprintf(stderr, "(width * height) %% 2 != 0, width = %d, height = %d.\n", w, h); return 1; }
        chroma = n / 2;
    } else if (base == "yuv422p") chroma = n;
    else if (base == "yuv444p") chroma = 2 * n;
    else { std::fprintf(stderr, "unknown format %s.\n", fmt.c_str()); return 1; }

    Plane ref, dis;
    ref.luma_bytes = dis.luma_bytes = n * bps;
    ref.chroma_bytes = dis.chroma_bytes = chroma * bps;
    if (!(ref.f = std::fopen(ref_path, "rb"))) { std::fprintf(stderr, "fopen ref_path %s failed.\n", ref_path); return 1; }
    const bool noref = app == "motion";                                // vmaf_main.c:58-90: motion reads the ref file only
    if (!noref && !(dis.f = std::fopen(dis_path, "rb"))) { std::fprintf(stderr, "fopen ref_path %s failed.\n", dis_path); return 1; }

    const int batch = (int)std::min<size_t>(64, std::max<size_t>(4, ((size_t)64 << 20) / (n * bps)));
    vmaf_hip_engine *e = nullptr;
    if (vmaf_hip_engine_create(&e, 0, w, h, batch)) { std::printf("error: %s\n", vmaf_hip_last_error()); return 1; }
    // this tool exists to print the raw atoms: the VIF plane sums in vif_sum_s's own order, so that its vif_* lines are the
    // reference tool's to the last printed digit (motion always is)
    if (vmaf_hip_engine_set_sums_exact(e, 1)) { std::printf("error: %s\n", vmaf_hip_last_error()); return 1; }
    if (do_ansnr && vmaf_hip_engine_enable_ansnr(e, bps == 2)) { std::printf("error: %s\n", vmaf_hip_last_error()); return 1; }
    char *rb = (char *)vmaf_hip_host_alloc(n * bps * batch), *db = (char *)vmaf_hip_host_alloc(n * bps * batch);
    std::vector<double> atoms((size_t)batch * VMAF_HIP_ATOMS), metrics((size_t)batch * VMAF_HIP_METRICS);
    if (!rb || !db) { std::printf("error: %s\n", vmaf_hip_last_error()); return 1; }

    int ret = 0, frm_idx = 0;
    for (bool eof = false; !eof && !ret;) {
        int k = 0;
        while (k < batch) {
            int r = read_luma(ref, rb + (size_t)k * n * bps);
            if (!r && !noref) r = read_luma(dis, db + (size_t)k * n * bps);
            if (r == 1) { std::printf("error: read frame failed.\n"); ret = 1; eof = true; break; }
            if (r == 2) { eof = true; break; }
            ++k;
        }
        if (!k || ret) break;
        if (vmaf_hip_extract_host_raw(e, rb, noref ? rb : db, bps, (size_t)w * bps, n * bps, k, frm_idx > 0, atoms.data())) {
            std::printf("error: %s\n", vmaf_hip_last_error());
            ret = 1;
            break;
        }
        if (do_ansnr && vmaf_hip_engine_get_metrics(e, metrics.data(), k)) { std::printf("error: %s\n", vmaf_hip_last_error()); ret = 1; break; }
        for (int i = 0; i < k; ++i, ++frm_idx) {
            const double *a = &atoms[(size_t)i * VMAF_HIP_ATOMS];
            if (do_adm) {                                              // adm.c:351-361 / all.c:138-146
                std::printf("adm: %d %f\n", frm_idx, a[1] == 0.0 ? 1.0 : a[0] / a[1]);   // adm.c:268
                std::printf("adm_num: %d %f\n", frm_idx, a[0]);
                std::printf("adm_den: %d %f\n", frm_idx, a[1]);
                for (int s = 0; s < 4; ++s) {
                    std::printf("adm_num_scale%d: %d %f\n", s, frm_idx, a[2 + 2 * s]);
                    std::printf("adm_den_scale%d: %d %f\n", s, frm_idx, a[3 + 2 * s]);
                }
            }
            if (do_ansnr) {                                            // ansnr.c:204-206 / all.c:173-176
                std::printf("ansnr: %d %f\n", frm_idx, metrics[(size_t)i * VMAF_HIP_METRICS + 3]);
                std::printf("anpsnr: %d %f\n", frm_idx, metrics[(size_t)i * VMAF_HIP_METRICS + 4]);
            }
            if (do_motion) std::printf("motion: %d %f\n", frm_idx, a[10]);   // motion.c:182 / all.c:204
            if (do_vif) {                                              // vif.c:373-382 / all.c:216-226
                double num = 0, den = 0;
                for (int s = 0; s < 4; ++s) { num += a[11 + 2 * s]; den += a[12 + 2 * s]; }
                std::printf("vif: %d %f\n", frm_idx, a[19]);
                std::printf("vif_num: %d %f\n", frm_idx, num);
                std::printf("vif_den: %d %f\n", frm_idx, den);
                for (int s = 0; s < 4; ++s) {
                    std::printf("vif_num_scale%d: %d %f\n", s, frm_idx, a[11 + 2 * s]);
                    std::printf("vif_den_scale%d: %d %f\n", s, frm_idx, a[12 + 2 * s]);
                }
            }
            std::fflush(stdout);
        }
    }
    vmaf_hip_host_free(rb); vmaf_hip_host_free(db);
    vmaf_hip_engine_destroy(e);
    if (ref.f) std::fclose(ref.f);
    if (dis.f) std::fclose(dis.f);
    return ret;
}
