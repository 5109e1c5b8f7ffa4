// raytracer — command-line front end over librt_mi355x.so, in the role of the reference's src/main.rs:
//   raytracer [-s N | --samples N] [-r | --random]
// reads ./scene.json (or builds the random scene with -r), renders on one MI355X and writes ./image.ppm.
// The extra options expose what the reference hard-codes: file names, resolution of the random scene,
// bounce depth, seeds, device.
#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_mi355x.h"

static void die(const char* what) {
    std::fprintf(stderr, "raytracer: %s: %s\n", what, rt_last_error());
    std::exit(101);  // the exit status of a Rust panic
}

static void usage(int code) {
    std::printf(
        "Usage: raytracer [OPTIONS]\n\n"
        "Options:\n"
        "  -s, --samples <SAMPLES_PER_PIXEL>  [default: 10]\n"
        "  -r, --random                       render the random-spheres scene instead of ./scene.json\n"
        "      --scene <PATH>                 scene file [default: ./scene.json]\n"
        "      --out <PATH>                   output image [default: ./image.ppm]\n"
        "      --width <W> --height <H>       resolution of the random scene [default: 512 512]\n"
        "      --depth <D>                    bounce limit [default: 10]\n"
        "      --seed <S>                     sampler seed [default: 0]\n"
        "      --scene-seed <S>               seed of the random scene [default: 1]\n"
        "      --device <N>                   GPU index [default: 0]\n"
        "      --gpus <G>                     split the samples over GPUs 0..G-1 of this node [default: 1]\n"
        "      --checkpoint <PATH>            keep the running sums in PATH: continue from it if it matches this\n"
        "                                     scene, frame, depth and seed, and save it after every pass\n"
        "      --every <N>                    samples per pass when --checkpoint is given [default: all]\n"
        "  -h, --help                         Print help\n"
        "  -V, --version                      Print version\n");
    std::exit(code);
}

int main(int argc, char** argv) {
    uint32_t spp = 10, width = 512, height = 512, depth = 10;  // src/main.rs:65, src/scenegen.rs:76-77, src/lib.rs:172
    uint64_t seed = 0, scene_seed = 1;
    bool random_scene = false;
    int device = 0;
    std::string scene_path = "./scene.json", out_path = "./image.ppm";  // src/main.rs:26, :38
    std::string checkpoint_path;
    uint32_t every = 0;
    int gpus = 1;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&]() -> const char* {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "-s" || a == "--samples") spp = uint32_t(std::strtoul(value(), nullptr, 10));
        else if (a.rfind("--samples=", 0) == 0) spp = uint32_t(std::strtoul(a.c_str() + 10, nullptr, 10));
        else if (a == "-r" || a == "--random") random_scene = true;
        else if (a == "--scene") scene_path = value();
        else if (a == "--out") out_path = value();
        else if (a == "--width") width = uint32_t(std::strtoul(value(), nullptr, 10));
        else if (a == "--height") height = uint32_t(std::strtoul(value(), nullptr, 10));
        else if (a == "--depth") depth = uint32_t(std::strtoul(value(), nullptr, 10));
        else if (a == "--seed") seed = std::strtoull(value(), nullptr, 10);
        else if (a == "--scene-seed") scene_seed = std::strtoull(value(), nullptr, 10);
        else if (a == "--device") device = std::atoi(value());
        else if (a == "--gpus") gpus = std::atoi(value());
        else if (a == "--checkpoint") checkpoint_path = value();
        else if (a == "--every") every = uint32_t(std::strtoul(value(), nullptr, 10));
        else if (a == "-h" || a == "--help") usage(0);
        else if (a == "-V" || a == "--version") { std::printf("raytracer 0.1.0 (MI355X backend, ABI %u.%u)\n", rt_abi_version() >> 16, rt_abi_version() & 0xffff); return 0; }
        else { std::fprintf(stderr, "error: unexpected argument '%s' found\n", a.c_str()); usage(2); }
    }

    rt_camera cam;
    std::vector<rt_hittable> generated;
    const rt_hittable* hittables = nullptr;
    const rt_texture* textures = nullptr;
    size_t n = 0, n_textures = 0;
    rt_config* cfg = nullptr;
    if (random_scene) {  // src/main.rs:22-24
        size_t count = 0;
        generated.resize(488);
        if (rt_gen_scene(scene_seed, generated.data(), generated.size(), &count)) die("gen_scene");
        generated.resize(count);
        std::printf("number of BoundingBox: %zu\n", count);  // src/scenegen.rs:65
        if (rt_default_camera(width, height, &cam)) die("default_camera");
        hittables = generated.data();
        n = count;
    } else {  // src/main.rs:25-33
        if (rt_config_load(scene_path.c_str(), &cfg)) die("scene");
        if (rt_config_view(cfg, &cam, &hittables, &n, &textures, &n_textures)) die("scene");
    }

    std::printf("Starting render...\n");
    std::printf("Computing with %llu samples\n", (unsigned long long)cam.horiz_res * cam.vert_res * spp);  // 64-bit, unlike src/main.rs:48
    std::fflush(stdout);
    const auto t0 = std::chrono::steady_clock::now();
    if (gpus > 1) {
        // one scene handle per device, the sample indices split among them, the sums added on the first device
        if (!checkpoint_path.empty()) { std::fprintf(stderr, "raytracer: --checkpoint and --gpus cannot be combined\n"); return 2; }
        std::vector<rt_scene*> scenes(size_t(gpus), nullptr);
        for (int g = 0; g < gpus; ++g)
            if (rt_scene_create(hittables, n, textures, n_textures, g, &scenes[size_t(g)])) die("scene");
        std::vector<uint8_t> rgb8(size_t(cam.horiz_res) * cam.vert_res * 3, 0);
        rt_render_opts o;
        std::memset(&o, 0, sizeof o);
        o.sample_end = spp;
        o.depth = depth;
        o.seed = seed;
        if (rt_render_multi(scenes.data(), scenes.size(), &cam, &o, nullptr, rgb8.data())) die("render");
        for (rt_scene* sc : scenes) rt_scene_destroy(sc);
        if (rt_write_ppm(out_path.c_str(), rgb8.data(), cam.horiz_res, cam.vert_res)) die("image");
    } else if (checkpoint_path.empty()) {
        // src/main.rs:35-42; unlike the reference the file is truncated, so a smaller image leaves no stale tail
        const int fd = ::open(out_path.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
        if (fd < 0) {
            std::fprintf(stderr, "raytracer: Unable to open file to write: %s: %s\n", out_path.c_str(), std::strerror(errno));
            return 101;
        }
        char header[64];
        const int hl = std::snprintf(header, sizeof header, "P3\n%u %u\n255\n", cam.horiz_res, cam.vert_res);
        if (::write(fd, header, size_t(hl)) != hl) {
            std::fprintf(stderr, "raytracer: Unable to write header to ppm\n");
            return 101;
        }
        const int rc = random_scene
                           ? rt_accel_render_into_file(fd, &cam, hittables, n, textures, n_textures, spp, depth, seed, device)
                           : rt_render_into_file(fd, &cam, hittables, n, textures, n_textures, spp, depth, seed, device);
        if (rc) die("render");
        ::close(fd);
    } else {
        // progressive render: passes of `every` samples, the running sums saved after each one
        const size_t n_px = size_t(cam.horiz_res) * cam.vert_res;
        std::vector<float> sum(n_px * 3, 0.0f);
        std::vector<uint8_t> rgb8(n_px * 3, 0);
        const uint64_t tag = rt_hash_bytes(hittables, n * sizeof(rt_hittable), rt_hash_bytes(&cam, sizeof cam, 0));
        rt_checkpoint_info info;
        uint32_t done = 0;
        if (rt_checkpoint_load(checkpoint_path.c_str(), &info, nullptr, 0) == RT_OK) {
            if (info.width != cam.horiz_res || info.height != cam.vert_res || info.depth != depth || info.seed != seed || info.tag != tag) {
                std::fprintf(stderr, "raytracer: %s belongs to another scene, frame, depth or seed\n", checkpoint_path.c_str());
                return 101;
            }
            if (rt_checkpoint_load(checkpoint_path.c_str(), &info, sum.data(), sum.size())) die("checkpoint");
            done = info.samples_done;
            std::printf("Continuing from %s: %u samples per pixel done\n", checkpoint_path.c_str(), done);
        }
        rt_scene* scene = nullptr;
        if (rt_scene_create(hittables, n, textures, n_textures, device, &scene)) die("scene");
        const uint32_t chunk = every ? every : (spp > done ? spp - done : 1);
        if (done >= spp && rt_finalize(sum.data(), n_px, done, rgb8.data())) die("image");  // nothing left to trace
        while (done < spp) {
            rt_render_opts o;
            std::memset(&o, 0, sizeof o);
            o.sample_begin = done;
            o.sample_end = std::min(spp, done + chunk);
            o.depth = depth;
            o.seed = seed;
            o.flags = RT_RENDER_ACCUMULATE;
            o.samples_per_item = every ? every : 0;  // aligned chunks: the passes add up exactly like one render in chunks of `every`
            if (rt_render(scene, &cam, &o, sum.data(), rgb8.data())) die("render");
            done = o.sample_end;
            info.width = cam.horiz_res; info.height = cam.vert_res; info.samples_done = done; info.depth = depth; info.seed = seed; info.tag = tag;
            if (rt_checkpoint_save(checkpoint_path.c_str(), &info, sum.data())) die("checkpoint");
            std::printf("%u of %u samples per pixel\n", done, spp);
            std::fflush(stdout);
        }
        rt_scene_destroy(scene);
        if (rt_write_ppm(out_path.c_str(), rgb8.data(), cam.horiz_res, cam.vert_res)) die("image");
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Render finished in %llus\n", (unsigned long long)s);  // src/main.rs:59
    rt_config_destroy(cfg);
    return 0;
}
