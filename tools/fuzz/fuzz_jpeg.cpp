#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>
#include "../../include/rt_mi355x.h"
int main(int argc, char** argv) {
    int ok = 0, err = 0;
    for (int f = 1; f < argc; ++f) {
        FILE* fp = fopen(argv[f], "rb"); if (!fp) return 1;
        std::vector<uint8_t> base; uint8_t buf[4096]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) base.insert(base.end(), buf, buf + n);
        fclose(fp);
        uint64_t s = 12345 + f;
        auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return uint32_t(s >> 33); };
        for (int t = 0; t < 4000; ++t) {
            std::vector<uint8_t> raw = base;
            int k = 1 + rnd() % 8;
            for (int i = 0; i < k; ++i) raw[2 + rnd() % (raw.size() - 2)] = uint8_t(rnd());
            if (t % 4 == 0) raw.resize(4 + rnd() % (raw.size() - 4));
            uint8_t* rgb = nullptr; uint32_t w = 0, h = 0;
            if (rt_decode_jpeg(raw.data(), raw.size(), &rgb, &w, &h) == 0) { ok++; rt_free(rgb); } else err++;
        }
    }
    printf("ok %d err %d\n", ok, err);
}
