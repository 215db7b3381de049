// ThreadSanitizer run of the front-end's threaded paths (round 6): the decoder's helper threads (zj_crew.h), restart segments on
// several threads and scan_baseline_parallel.  Built by tests/test_tsan.py from zj_jpeg.cpp + tests/fuzz/jpeg_stubs.cpp with
// -fsanitize=thread; decodes every file given on the command line with 1, 2, 3, 4 and 7 threads, several times per decoder
// (the crew is reused) and with fresh decoders, and compares the planes with the one-thread decode.  Exit 0: equal and no
// report (TSAN_OPTIONS=halt_on_error=1 turns a report into a non-zero exit).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/zjhip.h"

static std::vector<uint8_t> slurp(const char* path)
{
    std::vector<uint8_t> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

struct Planes { int rc; std::vector<int16_t> p[3]; };

static Planes decode(zj_decoder* d, const std::vector<uint8_t>& file)
{
    Planes out;
    zj_frame_desc fd;
    zj_image_info info;
    const int16_t* planes[3] = {nullptr, nullptr, nullptr};
    size_t len[3] = {0, 0, 0};
    out.rc = zj_decoder_decode_coefficients(d, file.data(), file.size(), &fd, planes, len, &info);
    if (out.rc == ZJ_OK)
        for (int i = 0; i < 3; i++)
            if (planes[i]) out.p[i].assign(planes[i], planes[i] + len[i]);
    return out;
}

int main(int argc, char** argv)
{
    setenv("ZJ_PAR_MIN_CHUNK", "1024", 1);
    long long parallel = 0;
    for (int a = 1; a < argc; a++) {
        const std::vector<uint8_t> file = slurp(argv[a]);
        if (file.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        zj_options o;
        memset(&o, 0, sizeof o); // (zero = the reference's defaults)
        o.num_threads = 1;
        zj_decoder* d1 = zj_decoder_new(&o);
        const Planes ref = decode(d1, file);
        zj_decoder_free(d1);
        const int counts[] = {2, 3, 4, 7};
        for (int threads : counts) {
            o.num_threads = threads;
            zj_decoder* d = zj_decoder_new(&o);
            for (int rep = 0; rep < 3; rep++) {
                const Planes got = decode(d, file);
                parallel += zj_decoder_parallel_mcus(d) + zj_decoder_parallel_segments(d);
                if (got.rc != ref.rc || got.p[0] != ref.p[0] || got.p[1] != ref.p[1] || got.p[2] != ref.p[2]) {
                    fprintf(stderr, "%s: %d threads differ from one (rc %d vs %d)\n", argv[a], threads, got.rc, ref.rc);
                    return 1;
                }
            }
            zj_decoder_free(d);
        }
    }
    printf("tsan harness: %d files, %lld MCUs / segments decoded on several threads, equal to one thread\n", argc - 1, parallel);
    return parallel > 0 ? 0 : 3;
}
