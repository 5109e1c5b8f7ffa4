#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/rt_mi355x.h"
extern "C" {
int rt_scene_create(const rt_hittable*, size_t, const rt_texture*, size_t, int, rt_scene**) { return 3; }
void rt_scene_destroy(rt_scene*) {}
int rt_render(rt_scene*, const rt_camera*, const rt_render_opts*, float*, uint8_t*) { return 3; }
}
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb"); if (!fp) return 1;
    std::string base; char buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) base.append(buf, n);
    fclose(fp);
    uint64_t s = 777;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return uint32_t(s >> 33); };
    const char alphabet[] = "{}[]\":,0123456789.eE+-\\u tfn\n";
    int ok = 0, err = 0;
    for (int t = 0; t < 20000; ++t) {
        std::string raw = base;
        int k = 1 + rnd() % 4;
        for (int i = 0; i < k; ++i) {
            size_t pos = rnd() % raw.size();
            switch (rnd() % 4) {
                case 0: raw[pos] = alphabet[rnd() % (sizeof alphabet - 1)]; break;
                case 1: raw.erase(pos, 1 + rnd() % 8); break;
                case 2: raw.insert(pos, 1, alphabet[rnd() % (sizeof alphabet - 1)]); break;
                default: raw[pos] = char(rnd());
            }
            if (raw.empty()) raw = "{";
        }
        if (t % 7 == 0) raw.resize(rnd() % (raw.size() + 1));
        rt_config* cfg = nullptr;
        if (rt_config_parse(raw.data(), raw.size(), argv[2], &cfg) == 0) { ok++; rt_config_destroy(cfg); } else err++;
    }
    printf("ok %d err %d\n", ok, err);
}
