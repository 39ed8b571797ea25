// main.cc -- command line front end `rrt`, keeping the reference's CLI (src/main.rs:28-55, README usage):
//
//   rrt [-t N] [-o FILE] [-g FILE] [-a RATIO] <HEIGHT> <SAMPLES>
//
// Defaults as in the reference: --gltf assets/default.gltf, --output output/render.png, camera and lights
// from assets/suzanne.gltf (src/gltf.rs:317), MAX_DEPTH 4, t_min 1e-4.  Extensions: --seed, --device, --gpus N,
// --devices LIST, --camera-lights FILE, --extended [--scale S], --gpu-build, --stats.
// --gpus N renders on devices 0..N-1 of this node in one process (rrt_group_*): the sample range is split over the
// GPUs and the accumulation buffers are summed on the first one (RCCL reduce over xGMI).
// All rendering happens on the GPU through the C ABI in include/rrt_ffi.h.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rrt_ffi.h"

namespace {

void usage() {
    std::fprintf(stderr,
                 "Ray Tracer 0.1 (MI355X)\nRay-tracing based rendering engine\n\n"
                 "USAGE:\n    rrt [OPTIONS] <HEIGHT> <SAMPLES>\n\n"
                 "OPTIONS:\n"
                 "    -a, --aspect_ratio <FLOAT>     Sets the camera aspect ratio\n"
                 "    -g, --gltf <FILE>              Sets the input glTF scene file\n"
                 "    -o, --output <FILE>            Sets the output image file name\n"
                 "    -t, --threads <NUM_THREADS>    Sets the number of host threads (BVH build)\n"
                 "        --seed <N>                 RNG seed (default 1)\n"
                 "        --device <N>               HIP device ordinal (default 0)\n"
                 "        --gpus <N>                 render on GPUs 0..N-1: samples split per GPU, RCCL reduce to GPU 0\n"
                 "        --devices <A,B,...>        the same with an explicit device list (a device may repeat)\n"
                 "        --camera-lights <FILE>     glTF file providing camera and lights (default assets/suzanne.gltf)\n"
                 "        --extended                 extended glTF loader (external .bin, node transforms, u32 indices)\n"
                 "        --scale <FLOAT>            uniform scale for --extended\n"
                 "        --gpu-build                build large BVHs on the GPU (faster build, slower render)\n"
                 "        --stats                    print a JSON line with rays, samples and kernel time\n\n"
                 "ARGS:\n    <HEIGHT>     Sets the image height\n    <SAMPLES>    Sets the number of samples per pixel\n");
}

int die(const char* what, const char* msg) {
    std::fprintf(stderr, "error: %s: %s\n", what, msg ? msg : "");
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    std::string gltf = "assets/default.gltf", output = "output/render.png", cam_lights = "assets/suzanne.gltf";
    std::string aspect_arg, threads_arg;
    std::vector<std::string> positional;
    uint32_t seed = 1;
    int device = 0;
    std::vector<int32_t> devices;  // empty: one context on `device`
    bool extended = false, stats_json = false, gpu_build = false;
    float ext_scale = 1.0f;

    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&](const char* long_name, const char* short_name, std::string& dst) -> bool {
            const std::string ln = std::string("--") + long_name;
            if (a == ln || (short_name && a == short_name)) {
                if (i + 1 >= argc) { std::fprintf(stderr, "error: %s needs a value\n", a.c_str()); std::exit(2); }
                dst = argv[++i];
                return true;
            }
            if (a.compare(0, ln.size() + 1, ln + "=") == 0) { dst = a.substr(ln.size() + 1); return true; }
            if (short_name && a.size() > 2 && a.compare(0, 2, short_name) == 0) { dst = a.substr(a[2] == '=' ? 3 : 2); return true; }
            return false;
        };
        std::string tmp;
        if (a == "-h" || a == "--help") { usage(); return 0; }
        if (a == "-V" || a == "--version") { std::printf("Ray Tracer 0.1\n"); return 0; }
        if (value("threads", "-t", threads_arg)) continue;
        if (value("output", "-o", output)) continue;
        if (value("gltf", "-g", gltf)) continue;
        if (value("aspect_ratio", "-a", aspect_arg)) continue;
        if (value("seed", nullptr, tmp)) { seed = (uint32_t)std::strtoul(tmp.c_str(), nullptr, 10); continue; }
        if (value("device", nullptr, tmp)) { device = std::atoi(tmp.c_str()); continue; }
        if (value("gpus", nullptr, tmp)) {
            const int n = std::atoi(tmp.c_str());
            if (n < 1 || n > 64) return die("--gpus", "1..64");
            devices.clear();
            for (int k = 0; k < n; ++k) devices.push_back(k);
            continue;
        }
        if (value("devices", nullptr, tmp)) {
            devices.clear();
            size_t pos = 0;
            while (pos <= tmp.size()) {
                const size_t comma = std::min(tmp.find(',', pos), tmp.size());
                if (comma == pos) return die("--devices", "empty entry in the list");
                devices.push_back((int32_t)std::atoi(tmp.substr(pos, comma - pos).c_str()));
                pos = comma + 1;
            }
            if (devices.empty() || devices.size() > 64) return die("--devices", "1..64 devices");
            continue;
        }
        if (value("camera-lights", nullptr, cam_lights)) continue;
        if (value("scale", nullptr, tmp)) { ext_scale = (float)std::atof(tmp.c_str()); continue; }
        if (a == "--extended") { extended = true; continue; }
        if (a == "--gpu-build") { gpu_build = true; continue; }
        if (a == "--stats") { stats_json = true; continue; }
        if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) { std::fprintf(stderr, "error: unknown option %s\n", a.c_str()); usage(); return 2; }
        positional.push_back(a);
    }
    if (positional.size() != 2) { usage(); return 2; }
    char* end = nullptr;
    const unsigned long height = std::strtoul(positional[0].c_str(), &end, 10);
    if (*end || height == 0) return die("HEIGHT", "not a positive integer");
    const unsigned long samples = std::strtoul(positional[1].c_str(), &end, 10);
    if (*end || samples == 0) return die("SAMPLES", "not a positive integer");
    if (!threads_arg.empty()) {
        const int t = std::atoi(threads_arg.c_str());
        if (t > 0) setenv("RRT_BUILD_THREADS", std::to_string(t).c_str(), 1);
    }

    rrt_gltf_options opt;
    std::memset(&opt, 0, sizeof opt);
    opt.flags = extended ? RRT_GLTF_EXTENDED : RRT_GLTF_COMPAT;
    opt.extended_scale = ext_scale;
    rrt_scene_desc* scene = nullptr;
    if (rrt_gltf_load(gltf.c_str(), cam_lights.c_str(), &opt, &scene) != RRT_OK) return die("scene", rrt_last_error(nullptr));

    // main.rs:76-81: the CLI aspect ratio only changes the image width, not the camera
    float aspect = scene->camera.aspect;
    if (!aspect_arg.empty()) aspect = (float)std::atof(aspect_arg.c_str());
    const float wf = (float)height * aspect;
    const uint32_t width = wf >= 4294967296.0f ? 0xFFFFFFFFu : (wf > 0.0f ? (uint32_t)wf : 0u);
    if (width == 0) return die("aspect_ratio", "image width is zero");

    rrt_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.device = device;
    cfg.seed = seed;
    cfg.max_depth = 4;   // main.rs:56
    cfg.t_min = 0.0001f; // ray.rs:89
    cfg.flags = gpu_build ? RRT_CFG_DEVICE_BUILD : 0u;
    std::vector<float> accum((size_t)width * height * 3);
    std::vector<uint8_t> rgb((size_t)width * height * 3);
    rrt_stats st;
    rrt_scene_info info;
    double wall = 0.0;
    const char* reduce_path = "none";
    std::string reduce_path_copy;
    if (devices.empty()) {
        rrt_ctx* ctx = nullptr;
        if (rrt_create(&cfg, &ctx) != RRT_OK) return die("device", rrt_last_error(nullptr));
        if (rrt_scene_upload(ctx, scene) != RRT_OK) return die("scene upload", rrt_last_error(ctx));
        rrt_scene_free(scene);
        const auto t0 = std::chrono::steady_clock::now();
        if (rrt_render(ctx, width, (uint32_t)height, (uint32_t)samples, 0u, accum.data(), &st) != RRT_OK) return die("render", rrt_last_error(ctx));
        if (rrt_tonemap(ctx, accum.data(), width, (uint32_t)height, (uint32_t)samples, rgb.data()) != RRT_OK) return die("tonemap", rrt_last_error(ctx));
        wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        (void)rrt_scene_get_info(ctx, &info);
        rrt_destroy(ctx);
    } else {
        rrt_group* grp = nullptr;
        if (rrt_group_create(&cfg, devices.data(), (uint32_t)devices.size(), &grp) != RRT_OK) return die("devices", rrt_group_last_error(nullptr));
        if (rrt_group_scene_upload(grp, scene) != RRT_OK) return die("scene upload", rrt_group_last_error(grp));
        rrt_scene_free(scene);
        const auto t0 = std::chrono::steady_clock::now();
        if (rrt_group_render(grp, width, (uint32_t)height, (uint32_t)samples, 0u, accum.data(), &st) != RRT_OK) return die("render", rrt_group_last_error(grp));
        if (rrt_group_tonemap(grp, accum.data(), width, (uint32_t)height, (uint32_t)samples, rgb.data()) != RRT_OK) return die("tonemap", rrt_group_last_error(grp));
        wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        (void)rrt_group_scene_get_info(grp, &info);
        reduce_path_copy = rrt_group_reduce_path(grp);
        reduce_path = reduce_path_copy.c_str();
        rrt_group_destroy(grp);
    }
    if (stats_json)
        std::printf("{\"width\": %u, \"height\": %lu, \"spp\": %lu, \"gpus\": %zu, \"reduce\": \"%s\", \"rays\": %llu, \"samples\": %llu, "
                    "\"kernel_ms\": %.3f, \"wall_s\": %.3f, \"mrays_per_s\": %.1f, \"prims\": %u, \"bvh_nodes\": %u, \"bvh_build_ms\": %.1f}\n",
                    width, height, samples, devices.empty() ? (size_t)1 : devices.size(), reduce_path, (unsigned long long)st.rays,
                    (unsigned long long)st.samples, st.kernel_ms, wall, st.kernel_ms > 0 ? (double)st.rays / st.kernel_ms * 1e-3 : 0.0,
                    info.prim_count, info.node_count, info.build_ms);
    if (rrt_png_write(output.c_str(), rgb.data(), width, (uint32_t)height) != RRT_OK) return die("output", rrt_last_error(nullptr));
    return 0;
}
