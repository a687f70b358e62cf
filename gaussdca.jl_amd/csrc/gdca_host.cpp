// Host-side utilities around the hot path (SURVEY.md 8f "next" rows f-1..f-3): the FASTA(.gz) reader
// with gap-fraction filter and letter map (DCAUtils.read_fasta_alignment; reference call site
// src/GaussDCA.jl:20), duplicate removal (DCAUtils.remove_duplicate_sequences; :21-23), the ranking
// enumeration + stable sort (compute_ranking; :88-99) and the "%i %i %e" writer (printrank; :67-74).
// Plain C++ (no HIP): these are host code in the reference too; they live in libgdca.so so that an
// end-to-end gDCA(filename) spends its time on the GPU, not in an interpreter loop.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <type_traits>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "gdca.h"
#include "gdca_inflate.h"

// The parsed matrix lives in a buffer that is NOT zero-filled (the reader overwrites every byte it keeps; zero-filling 50 MB per
// family is a serial 10 ms) and that is REUSED: gdca_fasta_close hands it to a small process-wide pool, the next gdca_fasta_open
// takes one that is large enough.  A batch driver parses hundreds of families on dozens of threads; as malloc / free of 2 MiB-
// aligned, huge-page-hinted blocks every matrix was an mmap + thousands of page faults + an munmap, all serialised on the
// process's memory-map lock -- that lock, not parsing, was what the parser threads queued on (profiles/r03_parse_bench.log:
// slower at 64 threads than at 32).  The pool holds at most POOL_SLOTS buffers and POOL_BYTES bytes; what does not fit is freed.
namespace {
struct MatBuf {
    int8_t *p = nullptr;
    size_t cap = 0;
};
constexpr size_t POOL_SLOTS = 96, POOL_BYTES = (size_t)6 << 30;
std::mutex g_pool_mu;
struct Pool : std::vector<MatBuf> {
    ~Pool()  // process exit: the pooled buffers go back (nothing is left for a leak checker to find)
    {
        for (MatBuf &b : *this) free(b.p);
    }
};
Pool g_pool;
size_t g_pool_bytes = 0;

MatBuf matbuf_get(size_t bytes)
{
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        // best fit among the pooled buffers that are large enough (and not absurdly larger: a 100 MB buffer is not spent on 1 MB)
        size_t best = g_pool.size();
        for (size_t k = 0; k < g_pool.size(); ++k)
            if (g_pool[k].cap >= bytes && g_pool[k].cap <= 8 * bytes + ((size_t)4 << 20) && (best == g_pool.size() || g_pool[k].cap < g_pool[best].cap))
                best = k;
        if (best != g_pool.size()) {
            MatBuf b = g_pool[best];
            g_pool[best] = g_pool.back();
            g_pool.pop_back();
            g_pool_bytes -= b.cap;
            return b;
        }
    }
    MatBuf b;
    if (bytes >= ((size_t)4 << 20)) {
        const size_t len = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void *q = nullptr;
        if (posix_memalign(&q, (size_t)2 << 20, len) != 0) return b;
        (void)madvise(q, len, MADV_HUGEPAGE);
        b.p = static_cast<int8_t *>(q);
        b.cap = len;
    } else {
        b.p = static_cast<int8_t *>(malloc(bytes));
        b.cap = b.p ? bytes : 0;
    }
    return b;
}

void matbuf_put(MatBuf b)
{
    if (!b.p) return;
    if (b.cap >= ((size_t)1 << 20)) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool.size() < POOL_SLOTS && g_pool_bytes + b.cap <= POOL_BYTES) {
            g_pool.push_back(b);
            g_pool_bytes += b.cap;
            return;
        }
    }
    free(b.p);
}
}  // namespace

struct gdca_fasta {
    int32_t N = 0, M = 0, qmax = 0;
    MatBuf Z;  // [M][N]
    ~gdca_fasta() { matbuf_put(Z); }
};

namespace {

// ACDEFGHIKLMNPQRSTVWY -> 1..20, everything else -> 21
struct LetterMap {
    int8_t t[256];
    LetterMap()
    {
        for (int i = 0; i < 256; ++i) t[i] = 21;
        const char *L = "ACDEFGHIKLMNPQRSTVWY";
        for (int i = 0; L[i]; ++i) t[(unsigned char)L[i]] = (int8_t)(i + 1);
    }
};
const LetterMap kMap;

// insert columns of an aligned FASTA record: '.' and lowercase letters (dropped by the reader)
struct InsertMap {
    uint8_t t[256];
    InsertMap()
    {
        for (int i = 0; i < 256; ++i) t[i] = (i == '.' || (i >= 'a' && i <= 'z')) ? 1 : 0;
    }
};
const InsertMap kIns;

// threads one gdca_fasta_open call may use on a big file: GDCA_FASTA_THREADS, default min(16, hardware threads).  The batch
// driver, which already parses several files at once, sets it to hardware threads / parser threads.
// CPUs this process can really use: the hardware threads, capped by the cgroup's CFS quota (a container that sees 256 threads
// but is given "1600000 100000" in cpu.max runs 16 of them at a time and is THROTTLED beyond that: on the GPU box every host
// loop of this project was fastest at 16 threads and slower at 32, 64, 128 for exactly this reason)
unsigned effective_cpus()
{
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    auto from = [&](const char *path_quota, const char *path_period) {
        FILE *f = fopen(path_quota, "r");
        if (!f) return;
        long long q = -1, per = 100000;
        char word[64] = {0};
        if (path_period) {
            if (fscanf(f, "%lld", &q) != 1) q = -1;
            if (FILE *g = fopen(path_period, "r")) {
                if (fscanf(g, "%lld", &per) != 1) per = 100000;
                fclose(g);
            }
        } else if (fscanf(f, "%63s %lld", word, &per) == 2 && strcmp(word, "max") != 0) {
            q = atoll(word);
        }
        fclose(f);
        if (q > 0 && per > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (q + per - 1) / per));
    };
    from("/sys/fs/cgroup/cpu.max", nullptr);
    from("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    return hw;
}

int fasta_threads()
{
    const unsigned hw = effective_cpus();
    if (const char *e = getenv("GDCA_FASTA_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) return std::min(v, 64);
    }
    return (int)(hw >= 64 ? std::min(32u, hw) : std::min(16u, hw));  // (25 MB of FASTA on a 256-thread host: 12.4 / 9 / 8.2 ms at 8 / 16 / 32 threads)
}

// The text of a file: a plain file is mapped (parsed where the page cache holds it: no copy, no buffer to fault in), a gzip
// file (magic 1f 8b) is inflated into a buffer the calling thread keeps between calls (a parser thread of the batch driver
// inflates hundreds of files: the buffer's pages are faulted in once).
struct FileText {
    std::string_view text;
    void *map = nullptr;
    size_t maplen = 0;
    ~FileText()
    {
        if (map) munmap(map, maplen);
    }
};

std::string &inflate_buffer()
{
    static thread_local std::string buf;
    return buf;
}

bool slurp(const char *path, FileText &out)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        close(fd);
        return false;
    }
    const size_t fsz = (size_t)sb.st_size;
    unsigned char magic[2] = {0, 0};
    const bool gz = fsz >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    std::string &buf = inflate_buffer();
    // the thread's buffer is kept between files, but not a huge one for ever: beyond 256 MiB and four times what this file needs
    // it goes back to the system (a batch with a few giant families would otherwise pin ~1 GiB per parser thread)
    const size_t need_guess = gz ? 8 * fsz : fsz;
    if (buf.capacity() > ((size_t)256 << 20) && buf.capacity() > 4 * need_guess) std::string().swap(buf);
    if (!gz) {
        if (fsz == 0) {
            close(fd);
            out.text = std::string_view();
            return true;
        }
        if (fasta_threads() == 1) {
            // one reader thread per file = a batch driver parsing many files at once: read() into the thread's own buffer (mapped
            // and faulted in once) instead of mapping every file -- an mmap / fault / munmap cycle per file takes the process's
            // memory-map lock three times, and with dozens of parser threads that lock was the bottleneck.  A file that shrinks
            // under the reader is a short read (GDCA_EINVAL), not a SIGBUS.
            if (buf.size() < fsz) buf.resize(fsz);
            size_t len = 0;
            while (len < fsz) {
                const ssize_t n = pread(fd, &buf[len], fsz - len, (off_t)len);
                if (n <= 0) break;
                len += (size_t)n;
            }
            close(fd);
            if (len != fsz) return false;
            out.text = std::string_view(buf.data(), len);
            return true;
        }
        void *m = mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        (void)madvise(m, fsz, MADV_SEQUENTIAL);
        out.map = m;
        out.maplen = fsz;
        out.text = std::string_view((const char *)m, fsz);
        return true;
    }
    // gzip: the trailer's ISIZE (uncompressed size mod 2^32) sizes the buffer up front -- as a HINT from an untrusted file: never
    // more than 64 times the compressed size (FASTA text deflates 3-10x; a crafted trailer must not make every parser thread
    // allocate 4 GiB), the buffer grows by doubling if the file really holds more
    size_t hint = 0;
    unsigned char t[4];
    if (fsz >= 4 && pread(fd, t, 4, (off_t)fsz - 4) == 4) hint = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
    hint = std::min(hint, 64 * fsz + ((size_t)1 << 16));
    // the compressed bytes into a second per-thread buffer, then zlib's inflate() straight into the text buffer (gzip members one
    // after the other, as gzread would): no gz* stream layer with its own megabyte buffers allocated and freed per file
    static thread_local std::string zin;
    if (zin.capacity() > ((size_t)256 << 20) && zin.capacity() > 4 * fsz) std::string().swap(zin);
    if (zin.size() < fsz + GDCA_INFLATE_PAD) zin.resize(fsz + GDCA_INFLATE_PAD);
    size_t got = 0;
    while (got < fsz) {
        const ssize_t n = pread(fd, &zin[got], fsz - got, (off_t)got);
        if (n <= 0) break;
        got += (size_t)n;
    }
    close(fd);
    if (got != fsz) return false;
    // the project's own decoder first (gdca_inflate.cpp: 2-3x zlib's inflate on alignment text, CRC-32 and ISIZE verified); whatever
    // it does not accept is decoded again by zlib below, whose verdict stands.  GDCA_FASTA_ZLIB=1: zlib only (A/B measurements).
    {
        static const bool zlib_only = getenv("GDCA_FASTA_ZLIB") != nullptr;
        memset(&zin[fsz], 0, GDCA_INFLATE_PAD);
        size_t len = 0;
        // (a big single-member file read by a caller that gives this file several threads -- one gDCA(filename) call, not the batch
        // driver's one-thread-per-file parsers -- is decoded on those threads: speculative block starts, see gdca_inflate.cpp)
        static const bool no_parallel = getenv("GDCA_FASTA_SERIAL_INFLATE") != nullptr;
        const int T = fasta_threads();
        if (!zlib_only && !no_parallel && T > 1 && fsz >= ((size_t)4 << 20) &&
            gdca_gunzip_parallel((const uint8_t *)zin.data(), fsz, buf, &len, hint, T)) {
            out.text = std::string_view(buf.data(), len);
            return true;
        }
        if (!zlib_only && gdca_gunzip_fast((const uint8_t *)zin.data(), fsz, buf, &len, hint)) {
            out.text = std::string_view(buf.data(), len);
            return true;
        }
    }
    if (buf.size() < std::max<size_t>(hint, 1 << 16)) buf.resize(std::max<size_t>(hint, 1 << 16));
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    zs.next_in = (Bytef *)zin.data();
    size_t in_left = fsz, len = 0;
    bool ok = true;
    for (;;) {
        if (len == buf.size()) buf.resize(buf.size() * 2);
        zs.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30);
        zs.next_out = (Bytef *)&buf[len];
        zs.avail_out = (uInt)std::min<size_t>(buf.size() - len, 1u << 30);
        const uInt in0 = zs.avail_in, out0 = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        in_left -= in0 - zs.avail_in;
        len += out0 - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (in_left == 0) break;
            // another gzip member follows (concatenated .gz files are one stream to gzip -d and to gzread)
            Bytef *next = zs.next_in;
            if (inflateReset(&zs) != Z_OK) {
                ok = false;
                break;
            }
            zs.next_in = next;
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) {
            ok = false;
            break;
        }
        if (rc == Z_BUF_ERROR && in_left == 0 && zs.avail_out != 0) {
            ok = false;  // truncated stream
            break;
        }
    }
    inflateEnd(&zs);
    if (!ok) return false;
    out.text = std::string_view(buf.data(), len);
    return true;
}

// One record without insert columns: row[i] = letter map of q[i]; *ins |= any insert character ('.', lowercase); *ngaps = number
// of '-'; *mx = largest mapped symbol.  Three table look-ups per byte as scalar code (~2.5 clocks a byte: 25 MB of FASTA were 50 ms
// of one thread); with AVX2 the map is two 16-entry shuffles on the low nibble -- the letters are 0x41 .. 0x59, so (c & 0x1f)
// indexes a 32-entry table, valid where (c & 0xe0) == 0x40 -- 32 bytes per step.
void map_record_scalar(const unsigned char *q, int8_t *row, int32_t n, unsigned *ins, int *ngaps, int *mx)
{
    unsigned in = 0;
    int g = 0, m = 0;
    for (int32_t i = 0; i < n; ++i) {
        const unsigned char c = q[i];
        in |= kIns.t[c];
        g += (c == '-');
        const int8_t v = kMap.t[c];
        row[i] = v;
        m = std::max(m, (int)v);
    }
    *ins |= in;
    *ngaps += g;
    *mx = std::max(*mx, m);
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void map_record_avx2(const unsigned char *q, int8_t *row, int32_t n, unsigned *ins, int *ngaps, int *mx)
{
    // table index = c & 0x1f:  '@' A B C D E F G H I J K L M N O | P Q R S T U V W X Y Z [ \ ] ^ _
    const __m256i lo = _mm256_setr_epi8(21, 1, 21, 2, 3, 4, 5, 6, 7, 8, 21, 9, 10, 11, 12, 21, 21, 1, 21, 2, 3, 4, 5, 6, 7, 8, 21, 9, 10, 11, 12, 21);
    const __m256i hi = _mm256_setr_epi8(13, 14, 15, 16, 17, 21, 18, 19, 21, 20, 21, 21, 21, 21, 21, 21, 13, 14, 15, 16, 17, 21, 18, 19, 21, 20, 21, 21,
                                        21, 21, 21, 21);
    const __m256i k0f = _mm256_set1_epi8(0x0f), k10 = _mm256_set1_epi8(0x10), ke0 = _mm256_set1_epi8((char)0xe0), k40 = _mm256_set1_epi8(0x40),
                  k21 = _mm256_set1_epi8(21), kdash = _mm256_set1_epi8('-'), kdot = _mm256_set1_epi8('.'), ka = _mm256_set1_epi8('a'),
                  k25 = _mm256_set1_epi8(25);
    __m256i vmax = _mm256_setzero_si256(), vins = _mm256_setzero_si256();
    int g = 0;
    int32_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i c = _mm256_loadu_si256((const __m256i *)(q + i));
        const __m256i low = _mm256_and_si256(c, k0f);
        const __m256i sel_hi = _mm256_cmpeq_epi8(_mm256_and_si256(c, k10), k10);
        const __m256i r = _mm256_blendv_epi8(_mm256_shuffle_epi8(lo, low), _mm256_shuffle_epi8(hi, low), sel_hi);
        const __m256i upper = _mm256_cmpeq_epi8(_mm256_and_si256(c, ke0), k40);
        const __m256i out = _mm256_blendv_epi8(k21, r, upper);
        _mm256_storeu_si256((__m256i *)(row + i), out);
        vmax = _mm256_max_epu8(vmax, out);
        g += __builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(c, kdash)));
        const __m256i t = _mm256_sub_epi8(c, ka);                          // c - 'a' (mod 256): 0 .. 25 for lowercase
        const __m256i lower = _mm256_cmpeq_epi8(_mm256_min_epu8(t, k25), t);
        vins = _mm256_or_si256(vins, _mm256_or_si256(lower, _mm256_cmpeq_epi8(c, kdot)));
    }
    unsigned in = _mm256_testz_si256(vins, vins) ? 0u : 1u;
    alignas(32) unsigned char mm[32];
    _mm256_store_si256((__m256i *)mm, vmax);
    int m = 0;
    for (int k = 0; k < 32; ++k) m = std::max(m, (int)mm[k]);
    *ins |= in;
    *ngaps += g;
    *mx = std::max(*mx, m);
    if (i < n) map_record_scalar(q + i, row + i, n - i, ins, ngaps, mx);
}
#endif

inline void map_record(const unsigned char *q, int8_t *row, int32_t n, unsigned *ins, int *ngaps, int *mx)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && !getenv("GDCA_FASTA_SCALAR");
    if (avx2) return map_record_avx2(q, row, n, ins, ngaps, mx);
#endif
    map_record_scalar(q, row, n, ins, ngaps, mx);
}

inline bool is_space(char c)
{
    return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f';
}

struct Span {
    size_t a, b;  // [a, b)
};

// run fn(t) for t = 0..T-1 on T threads (inline when T == 1)
template <class F>
void parallel(int T, F fn)
{
    if (T <= 1) {
        fn(0);
        return;
    }
    // (nothing may leave a worker thread -- that would end the process -- and nothing is lost: what a thread threw, or a thread that
    // could not be started, comes out of this function as std::bad_alloc in the caller's thread, after every started thread has ended)
    std::vector<std::thread> th;
    std::atomic<bool> failed{false};
    auto guarded = [&](int t) {
        try {
            fn(t);
        } catch (...) {
            failed = true;
        }
    };
    try {
        th.reserve((size_t)T - 1);
        for (int t = 1; t < T; ++t) th.emplace_back(guarded, t);
        guarded(0);
    } catch (...) {
        failed = true;
    }
    for (auto &x : th) x.join();
    if (failed) throw std::bad_alloc();
}

// the stripped non-empty lines of text[a, b), concatenated (what the reference's reader hands back as one
// sequence); `single` is set when the body is exactly one line, in which case nothing is copied
std::string_view body_sequence(std::string_view text, Span body, std::string &scratch)
{
    std::string_view first;
    int pieces = 0;
    size_t pos = body.a;
    while (pos < body.b) {
        const char *nl = (const char *)memchr(text.data() + pos, '\n', body.b - pos);
        const size_t eol = nl ? (size_t)(nl - text.data()) : body.b;
        size_t a = pos, b = eol;
        while (a < b && is_space(text[a])) ++a;
        while (b > a && is_space(text[b - 1])) --b;
        if (b > a) {
            if (pieces == 0) {
                first = std::string_view(text.data() + a, b - a);
            } else {
                if (pieces == 1) scratch.assign(first.data(), first.size());
                scratch.append(text.data() + a, b - a);
            }
            ++pieces;
        }
        pos = eol + 1;
    }
    return pieces <= 1 ? first : std::string_view(scratch);
}

}  // namespace

extern "C" {

int32_t gdca_host_cpus(void)
{
    return (int32_t)effective_cpus();
}

gdca_status gdca_fasta_open(const char *path, double max_gap_fraction, gdca_fasta **out, int32_t *N, int32_t *M)
try {
    if (!path || !out || !N || !M) return GDCA_EINVAL;
    *out = nullptr;
    // GDCA_FASTA_TRACE=1: per-file phase times on stderr (debug aid for the feed-rate benchmark, tools/parse_bench.sh)
    const bool trace = getenv("GDCA_FASTA_TRACE") != nullptr;
    auto tick = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_open = trace ? tick() : 0.0;
    FileText file;
    if (!slurp(path, file)) return GDCA_EINVAL;
    const double t_read = trace ? tick() : 0.0;
    const std::string_view text = file.text;
    const size_t L = text.size();
    const int T = L < (1u << 20) ? 1 : fasta_threads();

    // ONE team of T threads for both passes (round 6: as two parallel regions the 2 x 15 thread starts and joins were 0.5 ms of a
    // 3.5-ms parse at 25 MB, and a thread now parses the records of the very bytes it has just scanned for headers):
    //   phase A  header lines (first non-space character '>') of the thread's chunk of whole lines;
    //   -- barrier; thread 0: the records' numbering, the first record's columns, the matrix --
    //   phase B  the thread's own records: letter map + gap-fraction filter, every record into its own row.
    std::vector<size_t> cut((size_t)T + 1, L);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        const size_t guess = L / (size_t)T * (size_t)t;
        const char *nl = guess < L ? (const char *)memchr(text.data() + guess, '\n', L - guess) : nullptr;
        cut[(size_t)t] = nl ? (size_t)(nl - text.data()) + 1 : L;
    }
    std::vector<std::vector<Span>> found((size_t)T);
    std::vector<size_t> first_rec((size_t)T + 1, 0);  // number of the first record whose header thread t found
    size_t R = 0;
    std::string first;
    std::vector<uint32_t> cols;
    int32_t n = 0;
    bool all_match = false;
    gdca_status serial_status = GDCA_OK;
    std::unique_ptr<gdca_fasta> hold;
    gdca_fasta *h = nullptr;
    std::vector<uint8_t> keep;
    std::vector<int> qmax_t((size_t)T, 0);  // largest symbol among the rows each thread keeps
    std::atomic<bool> misaligned{false};
    double t_p1 = 0.0, t_alloc = 0.0;
    // where record r's body ends: at the next header, which may be the next thread's first
    auto header_of = [&](size_t r, int t_hint) -> const Span & {
        int t = t_hint;
        while (r >= first_rec[(size_t)t + 1]) ++t;
        return found[(size_t)t][r - first_rec[(size_t)t]];
    };
    struct TeamBarrier {
        std::mutex mu;
        std::condition_variable cv;
        int waiting = 0, phase = 0, T;
        explicit TeamBarrier(int T_) : T(T_) {}
        void arrive()
        {
            std::unique_lock<std::mutex> lk(mu);
            const int ph = phase;
            if (++waiting == T) {
                waiting = 0;
                ++phase;
                cv.notify_all();
            } else {
                cv.wait(lk, [&] { return phase != ph; });
            }
        }
    } bar(T);
    std::atomic<bool> team_failed{false};  // an exception inside a phase: the thread still meets the others at the barriers
    auto team_body = [&](int t) {
        try {
            size_t pos = cut[(size_t)t];
            const size_t end = cut[(size_t)t + 1];
            auto &mine = found[(size_t)t];
            while (pos < end) {
                const char *nl = (const char *)memchr(text.data() + pos, '\n', L - pos);
                const size_t eol = nl ? (size_t)(nl - text.data()) : L;
                size_t a = pos;
                while (a < eol && is_space(text[a])) ++a;
                if (a < eol && text[a] == '>') mine.push_back({pos, eol});
                pos = eol + 1;
            }
        } catch (...) {
            team_failed = true;
        }
        bar.arrive();
        if (t == 0) {
            try {
                t_p1 = trace ? tick() : 0.0;
                for (int u = 0; u < T; ++u) first_rec[(size_t)u + 1] = first_rec[(size_t)u] + found[(size_t)u].size();
                R = first_rec[(size_t)T];
                if (R == 0 || team_failed) {
                    serial_status = team_failed ? GDCA_ENOMEM : GDCA_EINVAL;
                } else {
                    // the first sequence fixes the alignment columns: everything except '.' and lowercase letters
                    std::string scratch0;
                    const Span h0 = header_of(0, 0);
                    const size_t e0 = R > 1 ? header_of(1, 0).a : L;
                    first = std::string(body_sequence(text, Span{std::min(L, h0.b + 1), e0}, scratch0));
                    for (size_t p = 0; p < first.size(); ++p) {
                        const char c = first[p];
                        if (c != '.' && !(c >= 'a' && c <= 'z')) cols.push_back((uint32_t)p);
                    }
                    n = (int32_t)cols.size();
                    if (n <= 0) {
                        serial_status = GDCA_EINVAL;
                    } else {
                        hold.reset(new (std::nothrow) gdca_fasta());  // (freed on every way out but the last line, exceptions included)
                        h = hold.get();
                        if (h) {
                            h->N = n;
                            h->Z = matbuf_get((size_t)n * R);
                        }
                        if (!h || !h->Z.p) {
                            hold.reset();
                            h = nullptr;
                            serial_status = GDCA_ENOMEM;
                        } else {
                            keep.assign(R, 0);
                            all_match = cols.size() == first.size();  // the first record has no insert columns (the usual case)
                        }
                    }
                }
                t_alloc = trace ? tick() : 0.0;
            } catch (...) {
                serial_status = GDCA_ENOMEM;
            }
        }
        bar.arrive();
        if (serial_status != GDCA_OK) return;
        std::string scratch;
        const size_t r0 = first_rec[(size_t)t], r1 = first_rec[(size_t)t + 1];
        for (size_t r = r0; r < r1 && !misaligned.load(std::memory_order_relaxed); ++r) {
            const Span hd = found[(size_t)t][r - r0];
            const size_t body_end = r + 1 < R ? (r + 1 < r1 ? found[(size_t)t][r + 1 - r0].a : header_of(r + 1, t).a) : L;
            const std::string_view sq = body_sequence(text, Span{std::min(L, hd.b + 1), body_end}, scratch);
            if (sq.size() != first.size()) {
                misaligned = true;  // "inputs are not aligned"
                return;
            }
            int8_t *row = h->Z.p + r * (size_t)n;
            if (all_match) {
                // one pass over the record: letter map, gap count, largest symbol, and "inconsistent inputs" = any insert character
                unsigned ins = 0;
                int ngaps = 0, mx = 0;
                map_record((const unsigned char *)sq.data(), row, n, &ins, &ngaps, &mx);
                if (ins) {
                    misaligned = true;
                    return;
                }
                keep[r] = (double)ngaps / (double)n <= max_gap_fraction;
                if (keep[r]) qmax_t[(size_t)t] = std::max(qmax_t[(size_t)t], mx);
                continue;
            }
            // "inconsistent inputs": the match columns of every record (neither '.' nor lowercase) must be exactly
            // those of the first one
            if (r > 0) {
                size_t nm = 0;
                bool same = true;
                for (size_t p = 0; p < sq.size(); ++p) {
                    const char c = sq[p];
                    if (c != '.' && !(c >= 'a' && c <= 'z')) {
                        if (nm >= cols.size() || cols[nm] != (uint32_t)p) {
                            same = false;
                            break;
                        }
                        ++nm;
                    }
                }
                if (!same || nm != cols.size()) {
                    misaligned = true;
                    return;
                }
            }
            int ngaps = 0;
            for (int32_t i = 0; i < n; ++i) {
                const unsigned char c = (unsigned char)sq[cols[(size_t)i]];
                ngaps += (c == '-');
                row[i] = kMap.t[c];
            }
            keep[r] = (double)ngaps / (double)n <= max_gap_fraction;
            if (keep[r])
                for (int32_t i = 0; i < n; ++i) qmax_t[(size_t)t] = std::max(qmax_t[(size_t)t], (int)row[i]);
        }
    };
    {
        // (nothing may leave a worker thread, and a thread that could not be started must not leave the others waiting at a barrier:
        // the team is as large as the number of threads that did start, and a team that is not complete fails the parse)
        std::vector<std::thread> th;
        auto guarded = [&](int t) {
            try {
                team_body(t);
            } catch (...) {
                team_failed = true;
            }
        };
        int started = 1;
        try {
            th.reserve((size_t)T - 1);
            for (int t = 1; t < T; ++t) {
                th.emplace_back(guarded, t);
                ++started;
            }
        } catch (...) {
            team_failed = true;
        }
        {
            std::lock_guard<std::mutex> lk(bar.mu);
            bar.T = started;  // (thread 0 has not arrived yet: nobody can be past the first barrier)
        }
        guarded(0);
        for (auto &x : th) x.join();
    }
    if (team_failed) {
        hold.reset();
        return GDCA_ENOMEM;
    }
    if (serial_status != GDCA_OK) return serial_status;
    const double t_p2 = trace ? tick() : 0.0;
    if (misaligned) {
        hold.reset();
        return GDCA_EINVAL;
    }
    // pass 3: drop the filtered rows, order preserved
    size_t m = 0;
    for (size_t r = 0; r < R; ++r)
        if (keep[r]) {
            if (m != r) memcpy(h->Z.p + m * (size_t)n, h->Z.p + r * (size_t)n, (size_t)n);
            ++m;
        }
    h->M = (int32_t)m;
    for (int v : qmax_t) h->qmax = std::max(h->qmax, (int32_t)v);
    if (trace)
        fprintf(stderr, "fasta-trace %s bytes %zu threads %d read/inflate %.2f ms parse %.2f ms (headers %.2f, first record + buffers %.2f, records %.2f, compaction %.2f)\n",
                path, L, T, t_read - t_open, tick() - t_read, t_p1 - t_read, t_alloc - t_p1, t_p2 - t_alloc, tick() - t_p2);
    *out = hold.release();
    *N = h->N;
    *M = h->M;
    return GDCA_OK;
} catch (const std::bad_alloc &) {
    return GDCA_ENOMEM;  // (never an exception across the C boundary)
} catch (...) {
    return GDCA_EINVAL;
}

gdca_status gdca_fasta_copy(const gdca_fasta *h, int8_t *Z)
{
    if (!h || !Z) return GDCA_EINVAL;
    if (h->M > 0) memcpy(Z, h->Z.p, (size_t)h->M * (size_t)h->N);
    return GDCA_OK;
}

const int8_t *gdca_fasta_data(const gdca_fasta *h)
{
    return h ? h->Z.p : nullptr;
}

int32_t gdca_fasta_max_symbol(const gdca_fasta *h)
{
    return h ? h->qmax : 0;
}

gdca_status gdca_fasta_close(gdca_fasta *h)
{
    delete h;
    return GDCA_OK;
}

gdca_status gdca_remove_duplicates(const int8_t *Z, int32_t N, int32_t M, int8_t *Z_out, int32_t *keep_idx,
                                   int32_t *M_out)
try {
    if (!Z || !Z_out || !M_out || N < 1 || M < 0) return GDCA_EINVAL;
    // The set's keys are views of the DESTINATION rows: row k is first copied to slot m of Z_out (m <= k, so with
    // Z_out == Z only a row that has already been examined is overwritten), then a view of that slot is inserted and
    // the slot is kept only if the insert succeeded.  Slots below m are never rewritten, so every key stays valid --
    // also when Z_out aliases Z (keys that pointed into Z itself were destroyed by the in-place compaction).
    std::unordered_set<std::string_view> seen;
    seen.reserve((size_t)M * 2);
    int32_t m = 0;
    for (int32_t k = 0; k < M; ++k) {
        int8_t *slot = Z_out + (size_t)m * N;
        const int8_t *src = Z + (size_t)k * N;
        if (slot != src) memmove(slot, src, (size_t)N);
        if (seen.insert(std::string_view((const char *)slot, (size_t)N)).second) {
            if (keep_idx) keep_idx[m] = k + 1;  // 1-based, as Julia returns them
            ++m;
        }
    }
    *M_out = m;
    return GDCA_OK;
} catch (const std::bad_alloc &) {
    return GDCA_ENOMEM;  // (never an exception across the C boundary)
} catch (...) {
    return GDCA_EINVAL;
}

int64_t gdca_ranking_length(int32_t N, int32_t min_separation)
{
    if (N < 1 || min_separation < 1 || min_separation >= N) return 0;
    const int64_t d = (int64_t)N - min_separation;
    return d * (d + 1) / 2;
}

gdca_status gdca_ranking(const double *S, int32_t N, int32_t min_separation, int32_t *i_out, int32_t *j_out,
                         double *score_out)
try {
    if (!S || N < 1 || min_separation < 1) return GDCA_EINVAL;
    const int64_t len = gdca_ranking_length(N, min_separation);
    if (len == 0) return GDCA_OK;
    if (!i_out || !j_out || !score_out) return GDCA_EINVAL;
    // sort!(R, by = x -> x[3], rev = true): a stable sort by `isless` on the score, reversed.  Done as a stable LSD
    // radix sort on a 64-bit key that orders doubles the way isless does (-0.0 < 0.0, every NaN greatest), bits
    // flipped for the descending direction; equal scores keep their generation order (i ascending, then j).
    const size_t n = (size_t)len;
    // (scratch of the calling thread, kept between calls: six arrays of n entries are 4 MB at N = 500 -- as fresh vectors every
    // call paid for mapping and faulting them in, which cost more than the sort)
    static thread_local std::vector<uint64_t> key, key2;
    static thread_local std::vector<uint32_t> idx, idx2;
    static thread_local std::vector<int32_t> gi, gj;
    if (key.capacity() > 16 * n + ((size_t)1 << 22)) {  // a giant earlier call does not pin its scratch for ever
        std::vector<uint64_t>().swap(key);
        std::vector<uint64_t>().swap(key2);
        std::vector<uint32_t>().swap(idx);
        std::vector<uint32_t>().swap(idx2);
        std::vector<int32_t>().swap(gi);
        std::vector<int32_t>().swap(gj);
    }
    key.resize(n);
    key2.resize(n);
    idx.resize(n);
    idx2.resize(n);
    gi.resize(n);
    gj.resize(n);
    size_t t = 0;
    for (int32_t i = 1; i <= N - min_separation; ++i)
        for (int32_t j = i + min_separation; j <= N; ++j, ++t) {
            const double x = S[(size_t)(j - 1) + (size_t)(i - 1) * N];  // S[j, i], column-major
            uint64_t u;
            memcpy(&u, &x, 8);
            uint64_t asc = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
            if (x != x) asc = ~0ull;
            key[t] = ~asc;
            idx[t] = (uint32_t)t;
            gi[t] = i;
            gj[t] = j;
        }
    // Stable LSD radix sort, 11 bits a pass, six passes (2048 write streams stay in the caches; with 16 bits a pass the 65 536
    // streams did not).  One thread: N = 500 is 1.2 ms, N = 1000 6.2 ms on the GPU box's host with the scratch above kept between
    // calls (it was 2-3 ms and 10-15 ms with fresh vectors); a multi-threaded form (per-thread histograms, spinning barrier) was
    // measured on the same host at 2 .. 16 threads and bought nothing reliable (1.6-2.9 ms and 4-16 ms: `tools/rank_time.cpp`).
    constexpr int RB = 11, NB = 1 << RB, PASSES = 6;
    static_assert(PASSES % 2 == 0, "the sorted arrays end up in key / idx");
    uint32_t hist[NB];
    uint64_t *src_k = key.data(), *dst_k = key2.data();
    uint32_t *src_i = idx.data(), *dst_i = idx2.data();
    for (int pass = 0; pass < PASSES; ++pass) {
        const int sh = RB * pass;
        for (int b = 0; b < NB; ++b) hist[b] = 0;
        for (size_t e = 0; e < n; ++e) hist[(src_k[e] >> sh) & (NB - 1)]++;
        uint32_t run = 0;
        for (int b = 0; b < NB; ++b) {
            const uint32_t c = hist[b];
            hist[b] = run;
            run += c;
        }
        for (size_t e = 0; e < n; ++e) {
            const uint32_t pos = hist[(src_k[e] >> sh) & (NB - 1)]++;
            dst_k[pos] = src_k[e];
            dst_i[pos] = src_i[e];
        }
        std::swap(src_k, dst_k);
        std::swap(src_i, dst_i);
    }
    for (size_t e = 0; e < n; ++e) {
        const uint32_t g = idx[e];
        i_out[e] = gi[g];
        j_out[e] = gj[g];
        score_out[e] = S[(size_t)(gj[g] - 1) + (size_t)(gi[g] - 1) * N];
    }
    return GDCA_OK;
} catch (const std::bad_alloc &) {
    return GDCA_ENOMEM;  // (never an exception across the C boundary)
} catch (...) {
    return GDCA_EINVAL;
}

gdca_status gdca_write_rank(const char *path, const int32_t *i, const int32_t *j, const double *score, int64_t len)
try {
    if (!path || (len > 0 && (!i || !j || !score))) return GDCA_EINVAL;
    FILE *f = fopen(path, "w");
    if (!f) return GDCA_EINVAL;
    for (int64_t t = 0; t < len; ++t) fprintf(f, "%i %i %e\n", i[t], j[t], score[t]);
    fclose(f);
    return GDCA_OK;
} catch (const std::bad_alloc &) {
    return GDCA_ENOMEM;  // (never an exception across the C boundary)
} catch (...) {
    return GDCA_EINVAL;
}

// ---- synthetic families ----------------------------------------------------------------------
namespace {
struct SplitMix {
    uint64_t s;
    // stream (tag, idx) of a seed: the start state is itself a SplitMix64 output, so streams do not overlap in practice
    SplitMix(uint64_t seed, uint64_t tag, uint64_t idx) : s(seed ^ (tag << 56) ^ idx) { s = next(); }
    uint64_t next()
    {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};
// one draw per site: high 32 bits against the threshold, low 32 bits pick the replacement symbol 1..q-1
inline int8_t resample(uint64_t r, uint32_t thresh, uint32_t nsym, int8_t keep)
{
    return (uint32_t)(r >> 32) < thresh ? (int8_t)(1 + (((r & 0xFFFFFFFFull) * nsym) >> 32)) : keep;
}
}  // namespace

gdca_status gdca_synth_family(int32_t N, int32_t M, int32_t q, uint64_t seed, int8_t *Z)
{
    if (!Z || N < 1 || M < 1 || q < 2 || q > 31) return GDCA_EINVAL;
    static const uint32_t kMu[6] = {85899345u, 214748364u, 429496729u, 858993459u, 1288490188u, 2147483648u};
    const uint32_t nsym = (uint32_t)(q - 1);
    std::vector<int8_t> root((size_t)N);
    {
        SplitMix g(seed, 0, 0);
        for (int32_t i = 0; i < N; ++i) root[(size_t)i] = (int8_t)(1 + g.below(nsym));
    }
    const int32_t K = (M + 24) / 25;
    std::vector<int8_t> centres((size_t)K * N);
    for (int32_t c = 0; c < K; ++c) {
        SplitMix g(seed, 1, (uint64_t)c);
        for (int32_t i = 0; i < N; ++i) centres[(size_t)c * N + i] = resample(g.next(), 1u << 30, nsym, root[(size_t)i]);
    }
    const uint32_t maxlen = (uint32_t)std::max(2, N / 10);
    for (int32_t k = 0; k < M; ++k) {
        SplitMix g(seed, 2, (uint64_t)k);
        const int8_t *cen = centres.data() + (size_t)g.below((uint32_t)K) * N;
        const uint32_t thr = kMu[g.below(6)];
        int8_t *row = Z + (size_t)k * N;
        for (int32_t i = 0; i < N; ++i) row[i] = resample(g.next(), thr, nsym, cen[i]);
        const uint32_t nruns = g.below(4);
        for (uint32_t r = 0; r < nruns; ++r) {
            const uint32_t a = g.below((uint32_t)N);
            const uint32_t len = 1 + g.below(maxlen);
            for (uint32_t i = a; i < std::min((uint32_t)N, a + len); ++i) row[i] = (int8_t)q;
        }
    }
    return GDCA_OK;
}

gdca_status gdca_write_fasta(const char *path, const int8_t *Z, int32_t N, int32_t M)
try {
    if (!path || !Z || N < 1 || M < 0) return GDCA_EINVAL;
    static const char L[] = "?ACDEFGHIKLMNPQRSTVWY-";
    std::string out;
    out.reserve((size_t)M * ((size_t)N + 16));
    char hdr[32];
    for (int32_t k = 0; k < M; ++k) {
        out.append(hdr, (size_t)snprintf(hdr, sizeof hdr, ">s%d\n", k));
        for (int32_t i = 0; i < N; ++i) {
            const int8_t a = Z[(size_t)k * N + i];
            if (a < 1 || a > 21) return GDCA_EINVAL;
            out.push_back(L[a]);
        }
        out.push_back('\n');
    }
    const size_t plen = strlen(path);
    if (plen > 3 && strcmp(path + plen - 3, ".gz") == 0) {
        gzFile f = gzopen(path, "wb1");
        if (!f) return GDCA_EINVAL;
        size_t off = 0;
        while (off < out.size()) {
            const unsigned n = (unsigned)std::min<size_t>(out.size() - off, 1u << 30);
            if (gzwrite(f, out.data() + off, n) != (int)n) {
                gzclose(f);
                return GDCA_EINVAL;
            }
            off += n;
        }
        return gzclose(f) == Z_OK ? GDCA_OK : GDCA_EINVAL;
    }
    FILE *f = fopen(path, "wb");
    if (!f) return GDCA_EINVAL;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return (fclose(f) == 0 && ok) ? GDCA_OK : GDCA_EINVAL;
} catch (const std::bad_alloc &) {
    return GDCA_ENOMEM;  // (never an exception across the C boundary)
} catch (...) {
    return GDCA_EINVAL;
}

}  // extern "C"
