// gdca_gunzip_fast (csrc/gdca_inflate.cpp) against zlib: test infrastructure, built by tests/test_inflate.py (plain and with
// AddressSanitizer + UBSan).  Modes:
//   inflate_check files a.gz b.gz ...      every file: fast decoder == zlib (bytes), or both fail; prints MB/s of each
//   inflate_check fuzz SEED ROUNDS a.gz    random corruptions / truncations of a.gz: the fast decoder must not crash, and whenever it
//                                          accepts an input its output must be what zlib produces for the same bytes
//   inflate_check crc                      gdca_crc32 against zlib's crc32 on random buffers and splits
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "gdca_inflate.h"

static bool zlib_gunzip(const std::vector<uint8_t> &in, std::string &out)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    out.resize(std::max<size_t>(in.size() * 4, 1 << 16));
    zs.next_in = (Bytef *)in.data();
    size_t in_left = in.size(), len = 0;
    bool ok = true;
    for (;;) {
        if (len == out.size()) out.resize(out.size() * 2);
        zs.avail_in = (uInt)in_left;
        zs.next_out = (Bytef *)&out[len];
        zs.avail_out = (uInt)(out.size() - len);
        const uInt in0 = zs.avail_in, out0 = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        in_left -= in0 - zs.avail_in;
        len += out0 - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (in_left == 0) break;
            Bytef *next = zs.next_in;
            if (inflateReset(&zs) != Z_OK) { ok = false; break; }
            zs.next_in = next;
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { ok = false; break; }
        if (rc == Z_BUF_ERROR && in_left == 0 && zs.avail_out != 0) { ok = false; break; }
    }
    inflateEnd(&zs);
    out.resize(ok ? len : 0);
    return ok;
}

static bool fast_gunzip(const std::vector<uint8_t> &in, std::string &buf, size_t *len)
{
    // an exact-size copy + padding: reads beyond the padding are caught by AddressSanitizer
    std::vector<uint8_t> padded(in.size() + GDCA_INFLATE_PAD, 0);
    if (!in.empty()) memcpy(padded.data(), in.data(), in.size());
    return gdca_gunzip_fast(padded.data(), in.size(), buf, len, 0);
}

static std::vector<uint8_t> read_file(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) v.insert(v.end(), tmp, tmp + n);
    fclose(f);
    return v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    int bad = 0;
    if (mode == "files") {
        for (int a = 2; a < argc; ++a) {
            const std::vector<uint8_t> in = read_file(argv[a]);
            std::string zo, fo;
            // best of five each; the fast decoder once more into a fresh buffer with no size hint (it has to grow it while decoding:
            // several times for anything beyond 64 KB) -- every call must give the same verdict
            bool zok = false, fok = false;
            double tz = 1e30, tf = 1e30;
            size_t flen = 0;
            int verdicts = 0;
            for (int rep = 0; rep < 5; ++rep) {
                double t0 = now();
                zok = zlib_gunzip(in, zo);
                tz = std::min(tz, now() - t0);
                if (!fo.empty()) memset(&fo[0], 0x5a, fo.size());   // (nothing left over from an earlier file or repetition can pass for output)
                t0 = now();
                fok = fast_gunzip(in, fo, &flen);
                tf = std::min(tf, now() - t0);
                verdicts += fok;
            }
            {
                std::string fresh;
                size_t l2 = 0;
                const bool ok2 = fast_gunzip(in, fresh, &l2);
                verdicts += ok2;
                if (ok2 && (!fok || l2 != flen || memcmp(fresh.data(), fo.data(), l2) != 0)) verdicts = -100;
            }
            // the multi-threaded decoder on the same bytes: it may decline (small files, several members, a speculation that failed),
            // but whatever it accepts must be zlib's output
            {
                std::vector<uint8_t> padded(in.size() + GDCA_INFLATE_PAD, 0);
                if (!in.empty()) memcpy(padded.data(), in.data(), in.size());
                for (int threads : {2, 5, 16}) {
                    std::string po;
                    size_t pl = 0;
                    double t0 = now();
                    const bool pok = gdca_gunzip_parallel(padded.data(), in.size(), po, &pl, 0, threads);
                    const double tp = now() - t0;
                    if (pok && (!zok || pl != zo.size() || memcmp(po.data(), zo.data(), pl) != 0)) {
                        printf("%s: the parallel decoder (%d threads) accepted the file and decoded it differently  MISMATCH\n", argv[a], threads);
                        ++bad;
                    }
                    if (pok) printf("   parallel x%d: ok, %.0f MB/s\n", threads, pl / tp / 1e6);
                    else if (in.size() > (4u << 20)) printf("   parallel x%d: declined\n", threads);
                }
            }
            if (verdicts != 0 && verdicts != 6) {
                printf("%s: the fast decoder's verdict or output depends on the state of its output buffer  MISMATCH\n", argv[a]);
                ++bad;
            }
            const bool same = fok ? (zok && flen == zo.size() && memcmp(fo.data(), zo.data(), flen) == 0) : true;
            printf("%s: zlib %s (%zu bytes, %.0f MB/s), fast %s (%.0f MB/s)%s\n", argv[a], zok ? "ok" : "FAILS", zo.size(), zo.size() / tz / 1e6,
                   fok ? "ok" : "declines", fok ? flen / tf / 1e6 : 0.0, same ? "" : "  MISMATCH");
            if (!same) ++bad;
            if (zok && !fok) printf("   (note: zlib accepts what the fast decoder declines)\n");
        }
    } else if (mode == "fuzz") {
        const unsigned seed = (unsigned)atoi(argv[2]);
        const int rounds = atoi(argv[3]);
        const std::vector<uint8_t> orig = read_file(argv[4]);
        std::mt19937 rng(seed);
        int accepted = 0, declined = 0, paccepted = 0;
        std::string zo, fo;
        for (int r = 0; r < rounds; ++r) {
            std::vector<uint8_t> in = orig;
            const int kind = (int)(rng() % 5);
            if (kind == 0) {
                in.resize(rng() % (in.size() + 1));                              // truncation
            } else if (kind == 1) {
                for (int k = 0, n = 1 + (int)(rng() % 4); k < n; ++k) in[rng() % in.size()] ^= (uint8_t)(1u << (rng() % 8));   // bit flips
            } else if (kind == 2) {
                const size_t at = rng() % in.size(), n = 1 + rng() % 64;
                for (size_t k = at; k < std::min(in.size(), at + n); ++k) in[k] = (uint8_t)rng();   // a run of random bytes
            } else if (kind == 3) {
                const size_t at = rng() % std::min<size_t>(in.size(), 64);        // damage near the header / first block header
                in[at] = (uint8_t)rng();
            } else {
                in.insert(in.end(), orig.begin(), orig.begin() + (long)(rng() % orig.size()));   // a second, truncated member
            }
            size_t flen = 0;
            if (!fo.empty()) memset(&fo[0], 0x5a, fo.size());
            {   // the multi-threaded decoder must survive the same input, and agree with zlib whenever it accepts it
                std::vector<uint8_t> padded(in.size() + GDCA_INFLATE_PAD, 0);
                if (!in.empty()) memcpy(padded.data(), in.data(), in.size());
                std::string po;
                size_t pl = 0;
                if (gdca_gunzip_parallel(padded.data(), in.size(), po, &pl, 0, 3)) {
                    ++paccepted;
                    if (!zlib_gunzip(in, zo) || zo.size() != pl || memcmp(zo.data(), po.data(), pl) != 0) {
                        printf("round %d kind %d: the parallel decoder accepted an input zlib rejects or decodes differently\n", r, kind);
                        ++bad;
                    }
                }
            }
            const bool fok = fast_gunzip(in, fo, &flen);
            if (fok) {
                ++accepted;
                const bool zok = zlib_gunzip(in, zo);
                if (!zok || zo.size() != flen || memcmp(zo.data(), fo.data(), flen) != 0) {
                    printf("round %d kind %d: fast decoder accepted an input zlib %s\n", r, kind, zok ? "decodes differently" : "rejects");
                    ++bad;
                }
            } else {
                ++declined;
            }
        }
        printf("fuzz: %d rounds, %d accepted (all equal to zlib: %s), %d declined; parallel decoder accepted %d\n", rounds, accepted, bad ? "NO" : "yes", declined, paccepted);
    } else if (mode == "crc") {
        std::mt19937 rng(7);
        std::vector<uint8_t> v(1 << 20);
        for (auto &x : v) x = (uint8_t)rng();
        for (int r = 0; r < 2000; ++r) {
            const size_t a = rng() % 4096, n = rng() % (r < 1000 ? 300 : v.size() - 4096), cut = n ? rng() % n : 0;
            const uint32_t want = (uint32_t)crc32(0, v.data() + a, (uInt)n);
            const uint32_t got = gdca_crc32(gdca_crc32(0, v.data() + a, cut), v.data() + a + cut, n - cut);
            if (want != got) { ++bad; printf("crc mismatch at a=%zu n=%zu\n", a, n); }
            const uint32_t comb = gdca_crc32_combine(gdca_crc32(0, v.data() + a, cut), gdca_crc32(0, v.data() + a + cut, n - cut), n - cut);
            if (comb != want) { ++bad; printf("crc combine mismatch at a=%zu n=%zu cut=%zu\n", a, n, cut); }
        }
        double t0 = now();
        uint32_t c = 0;
        for (int k = 0; k < 64; ++k) c = gdca_crc32(c, v.data(), v.size());
        const double t1 = now() - t0;
        t0 = now();
        uint32_t z = 0;
        for (int k = 0; k < 64; ++k) z = (uint32_t)crc32(z, v.data(), (uInt)v.size());
        const double t2 = now() - t0;
        printf("crc: %s; gdca_crc32 %.0f MB/s, zlib crc32 %.0f MB/s\n", (bad == 0 && c == z) ? "equal" : "MISMATCH", 64.0 * v.size() / t1 / 1e6, 64.0 * v.size() / t2 / 1e6);
        if (c != z) ++bad;
    }
    return bad ? 1 : 0;
}
