// gzip / DEFLATE fast path of the FASTA reader: see gdca_inflate.h.  Formats: RFC 1951 (DEFLATE), RFC 1952 (gzip).
#include "gdca_inflate.h"

#include <algorithm>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

// ---- decode tables ------------------------------------------------------------------------------------------------
// entry: bits 0-4 ALL the bits the entry consumes (its code bits -- in a sub-table: those beyond the root index -- plus the extra
// bits of a length / distance: one shift takes both, the extra bits are read out of a copy of the buffer, off the critical
// path), 5-7 kind, 8-11 extra-bit count (kind BASE) or sub-table index bits (kind SUB), 12-15 code bits alone (where the extra
// bits start), 16-31 value (literal byte(s), base of a length / distance, first entry of a sub-table)
enum : uint32_t { K_LIT = 0, K_LIT2 = 1, K_BASE = 2, K_EOB = 3, K_SUB = 4, K_BAD = 5 };  // (K_LIT2: two literals, value = first | second << 8)
constexpr uint32_t entry(uint32_t bits, uint32_t kind, uint32_t extra, uint32_t value)
{
    return (bits + (kind == K_BASE ? extra : 0)) | (kind << 5) | (extra << 8) | (bits << 12) | (value << 16);
}
constexpr uint32_t e_total(uint32_t e) { return e & 31; }
constexpr uint32_t e_kind(uint32_t e) { return (e >> 5) & 7; }
constexpr uint32_t e_extra(uint32_t e) { return (e >> 8) & 15; }
constexpr uint32_t e_cbits(uint32_t e) { return (e >> 12) & 15; }
constexpr uint32_t KIND_MASK = 7u << 5;
constexpr int LIT_BITS = 11, DIST_BITS = 8, PRE_BITS = 7;
constexpr int LIT_CAP = (1 << LIT_BITS) + 288 * 16, DIST_CAP = (1 << DIST_BITS) + 32 * 128, PRE_CAP = 1 << PRE_BITS;

const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t dist_base[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

enum Alphabet { LITLEN, DIST, PRECODE };

inline uint32_t symbol_entry(Alphabet a, int sym, uint32_t bits)
{
    if (a == LITLEN) {
        if (sym < 256) return entry(bits, K_LIT, 0, (uint32_t)sym);
        if (sym == 256) return entry(bits, K_EOB, 0, 0);
        if (sym <= 285) return entry(bits, K_BASE, len_extra[sym - 257], len_base[sym - 257]);
        return entry(bits, K_BAD, 0, 0);
    }
    if (a == DIST) return sym < 30 ? entry(bits, K_BASE, dist_extra[sym], dist_base[sym]) : entry(bits, K_BAD, 0, 0);
    return entry(bits, K_LIT, 0, (uint32_t)sym);
}

inline uint32_t reverse_bits(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// canonical Huffman code of `lens[0 .. nsym)` (RFC 1951 3.2.2) -> table with `root` index bits (+ sub-tables behind it).
// false: over-subscribed, or incomplete in a way DEFLATE does not allow (then the caller leaves the file to zlib).
bool build_table(Alphabet a, const uint8_t *lens, int nsym, int root, uint32_t *T, int cap)
{
    int count[16] = {0};
    for (int s = 0; s < nsym; ++s) count[lens[s]]++;
    const int used = nsym - count[0];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    if (left > 0) {
        // incomplete: only "no distance codes at all" (a block of literals) and "one distance code, of one bit" are legal
        if (!(a == DIST && (used == 0 || (used == 1 && count[1] == 1)))) return false;
    }
    const int rsize = 1 << root;
    for (int i = 0; i < rsize; ++i) T[i] = entry(1, K_BAD, 0, 0);
    uint32_t next_code[16];
    {
        uint32_t code = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (code + (uint32_t)count[l - 1]) << 1;
            next_code[l] = code;
        }
    }
    // codes up to `root` bits fill the root table; for the longer ones first find, per root prefix, the longest code behind it
    uint8_t maxlen[1 << LIT_BITS];
    bool any_long = false;
    uint16_t rev_of[288];
    for (int s = 0; s < nsym; ++s) {
        const int l = lens[s];
        if (l == 0) continue;
        const uint32_t rev = reverse_bits(next_code[l]++, l);
        rev_of[s] = (uint16_t)rev;
        if (l <= root) {
            const uint32_t e = symbol_entry(a, s, (uint32_t)l);
            for (int i = (int)rev; i < rsize; i += 1 << l) T[i] = e;
        } else {
            if (!any_long) {
                memset(maxlen, 0, (size_t)rsize);
                any_long = true;
            }
            uint8_t &m = maxlen[rev & (uint32_t)(rsize - 1)];
            if (l > m) m = (uint8_t)l;
        }
    }
    if (!any_long) return true;
    int next = rsize;
    for (int s = 0; s < nsym; ++s) {
        const int l = lens[s];
        if (l <= root) continue;
        const uint32_t prefix = rev_of[s] & (uint32_t)(rsize - 1);
        const int sub_bits = maxlen[prefix] - root;
        if (e_kind(T[prefix]) != K_SUB) {
            const int size = 1 << sub_bits;
            if (next + size > cap) return false;
            T[prefix] = entry((uint32_t)root, K_SUB, (uint32_t)sub_bits, (uint32_t)next);
            for (int i = 0; i < size; ++i) T[next + i] = entry(1, K_BAD, 0, 0);
            next += size;
        }
        const int at = (int)(T[prefix] >> 16);
        const uint32_t e = symbol_entry(a, s, (uint32_t)(l - root));
        for (int i = (int)(rev_of[s] >> root); i < (1 << sub_bits); i += 1 << (l - root)) T[at + i] = e;
    }
    return true;
}

// Alignment text is mostly literals of 4-5 bits, and the decoder's speed is set by the chain look-up -> shift -> look-up: where the
// code FOLLOWING a literal is a literal too and both fit into the root index, the root entry delivers the pair.
void add_literal_pairs(uint32_t *T)
{
    constexpr int R = 1 << LIT_BITS;
    uint32_t root[R];
    memcpy(root, T, sizeof root);
    for (int i = 0; i < R; ++i) {
        const uint32_t e = root[i];
        if (e_kind(e) != K_LIT) continue;
        const int l1 = (int)e_cbits(e), avail = LIT_BITS - l1;
        const uint32_t e2 = root[i >> l1];  // (the unknown upper bits read as zero: fine for a code of at most `avail` bits)
        if (e_kind(e2) != K_LIT || (int)e_cbits(e2) > avail) continue;
        T[i] = entry((uint32_t)l1 + e_cbits(e2), K_LIT2, 0, (e >> 16) | ((e2 >> 16) << 8));
    }
}

struct FixedTables {
    uint32_t lit[LIT_CAP], dist[DIST_CAP];
    bool ok;
    FixedTables()
    {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        ok = build_table(LITLEN, l, 288, LIT_BITS, lit, LIT_CAP) && build_table(DIST, d, 32, DIST_BITS, dist, DIST_CAP);
        if (ok) add_literal_pairs(lit);
    }
};

// ---- CRC-32, slicing-by-16 -------------------------------------------------------------------------------------------
struct CrcTables {
    uint32_t t[16][256];
    CrcTables()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 16; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};

inline uint64_t load64(const uint8_t *p)
{
    uint64_t w;
    memcpy(&w, p, 8);
    return w;  // (x86-64 / little-endian hosts: the project's host code is built for those only)
}

struct Bits {
    const uint8_t *p, *lim;  // next byte to load; p may run up to lim = end + 8 (the caller's padding covers the loads)
    uint64_t buf = 0;
    int cnt = 0;
    inline void refill()
    {
        buf |= load64(p) << cnt;
        const int adv = (63 - cnt) >> 3;
        p += adv;
        cnt += adv * 8;
    }
    inline uint32_t take(int n)
    {
        const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
    // position of the first byte not consumed at all, after dropping the rest of a partly consumed one
    inline const uint8_t *byte_align()
    {
        const int drop = cnt & 7;
        buf >>= drop;
        cnt -= drop;
        const uint8_t *q = p - (cnt >> 3);
        buf = 0;
        cnt = 0;
        return q;
    }
};

// the header of a dynamic block (RFC 1951 3.2.7), the block-type bits already consumed: both tables built.  false: not a
// valid header (the parallel decoder's search for block starts relies on how much has to be right for `true`)
bool read_dynamic_header(Bits &b, uint32_t *lit, uint32_t *dist)
{
    b.refill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t pl[19] = {0};
    b.refill();
    for (int i = 0; i < hclen; ++i) {
        if (b.cnt < 3) b.refill();
        pl[order[i]] = (uint8_t)b.take(3);
    }
    uint32_t PT[PRE_CAP];
    if (!build_table(PRECODE, pl, 19, PRE_BITS, PT, PRE_CAP)) return false;
    uint8_t lens[288 + 32 + 140];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        if (b.p > b.lim) return false;
        b.refill();
        const uint32_t e = PT[b.buf & (PRE_CAP - 1)];
        if (e_kind(e) != K_LIT) return false;
        b.take((int)e_total(e));
        const int sym = (int)(e >> 16);
        if (sym < 16) {
            lens[i++] = (uint8_t)sym;
        } else if (sym == 16) {
            if (i == 0) return false;
            const int rep = 3 + (int)b.take(2);
            memset(lens + i, lens[i - 1], (size_t)rep);
            i += rep;
        } else {
            const int rep = sym == 17 ? 3 + (int)b.take(3) : 11 + (int)b.take(7);
            memset(lens + i, 0, (size_t)rep);
            i += rep;
        }
    }
    if (i != total || lens[256] == 0) return false;
    uint8_t ll[288] = {0}, dl[32] = {0};
    memcpy(ll, lens, (size_t)hlit);
    memcpy(dl, lens + hlit, (size_t)hdist);
    return build_table(LITLEN, ll, 288, LIT_BITS, lit, LIT_CAP) && build_table(DIST, dl, 32, DIST_BITS, dist, DIST_CAP);
}

}  // namespace

static uint32_t crc32_tables(uint32_t c, const uint8_t *p, size_t n)  // c: the running (inverted) state
{
    static const CrcTables T;
    while (n && ((uintptr_t)p & 7)) {
        c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
        --n;
    }
    while (n >= 16) {
        const uint64_t a = load64(p) ^ c, b = load64(p + 8);
        c = T.t[15][a & 0xff] ^ T.t[14][(a >> 8) & 0xff] ^ T.t[13][(a >> 16) & 0xff] ^ T.t[12][(a >> 24) & 0xff] ^ T.t[11][(a >> 32) & 0xff] ^
            T.t[10][(a >> 40) & 0xff] ^ T.t[9][(a >> 48) & 0xff] ^ T.t[8][a >> 56] ^ T.t[7][b & 0xff] ^ T.t[6][(b >> 8) & 0xff] ^
            T.t[5][(b >> 16) & 0xff] ^ T.t[4][(b >> 24) & 0xff] ^ T.t[3][(b >> 32) & 0xff] ^ T.t[2][(b >> 40) & 0xff] ^ T.t[1][(b >> 48) & 0xff] ^
            T.t[0][b >> 56];
        p += 16;
        n -= 16;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c;
}

#if defined(__x86_64__)
// Carry-less-multiply folding (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009):
// four 128-bit accumulators are folded forward 64 bytes at a time (multiplication by x^(512+-32) mod P in the bit-reflected
// domain), then into one, which then steps forward 16 bytes at a time.  The last reduction is not the paper's Barrett step: the
// 128-bit accumulator is congruent to the message read so far, so the table CRC of its 16 bytes (from a zero state) is the state
// to carry on from -- two constants pairs instead of four, and the tail bytes go through the same table code.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t c, const uint8_t *p, size_t n)
{
    if (n < 64 + 16) return crc32_tables(c, p, n);
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);  // fold by 64 bytes: (hi, lo)
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);  // fold by 16 bytes
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16)),
            x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    p += 64;
    n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00), a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00),
                      a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        n -= 64;
    }
#define GDCA_FOLD16(x, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k3k4, 0x11), _mm_clmulepi64_si128(x, k3k4, 0x00)), next)
    x1 = GDCA_FOLD16(x1, x2);
    x1 = GDCA_FOLD16(x1, x3);
    x1 = GDCA_FOLD16(x1, x4);
    while (n >= 16) {
        x1 = GDCA_FOLD16(x1, _mm_loadu_si128((const __m128i *)p));
        p += 16;
        n -= 16;
    }
#undef GDCA_FOLD16
    alignas(16) uint8_t acc[16];
    _mm_store_si128((__m128i *)acc, x1);
    return crc32_tables(crc32_tables(0, acc, 16), p, n);
}
#endif

// CRC-32 of A || B from those of A and B and the length of B: crc(A) is multiplied by x^(8 len B) modulo the polynomial (bit-reflected:
// bit 31 is x^0), the power by square-and-multiply over a table of x^(2^k)
namespace {
inline uint32_t mulmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
struct X2nTable {
    uint32_t t[32];
    X2nTable()
    {
        uint32_t p = 1u << 30;  // x^1
        t[0] = p;
        for (int k = 1; k < 32; ++k) t[k] = p = mulmodp(p, p);
    }
};
}  // namespace

uint32_t gdca_crc32_combine(uint32_t crc_a, uint32_t crc_b, size_t len_b)
{
    static const X2nTable X;
    uint32_t p = 1u << 31;  // x^0
    size_t n = len_b;
    unsigned k = 3;         // x^(2^3) = one byte
    while (n) {
        if (n & 1) p = mulmodp(X.t[k & 31], p);
        n >>= 1;
        ++k;
    }
    return mulmodp(p, crc_a) ^ crc_b;
}

uint32_t gdca_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
#if defined(__x86_64__)
    static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (clmul) return ~crc32_clmul(~crc, p, n);
#endif
    return ~crc32_tables(~crc, p, n);
}

bool gdca_gunzip_fast(const uint8_t *in, size_t n, std::string &outbuf, size_t *len_out, size_t hint)
{
    static const FixedTables fixed;
    if (!fixed.ok) return false;
    const uint8_t *const in_end = in + n;
    constexpr size_t SLACK = 320;  // a trip of the symbol loop writes at most 258 + 15 bytes beyond `out`
    if (outbuf.size() < std::max<size_t>(hint + SLACK, (size_t)1 << 16)) outbuf.resize(std::max<size_t>(hint + SLACK, (size_t)1 << 16));
    uint8_t *base = (uint8_t *)&outbuf[0];
    uint8_t *out = base, *out_end = base + outbuf.size();
    auto grow = [&](size_t want) -> bool {
        const size_t pos = (size_t)(out - base);
        size_t size = outbuf.size();
        while (size - pos < want) size *= 2;
        if (size > 1040 * n + ((size_t)1 << 20)) return false;  // (DEFLATE cannot expand by more than 1032:1)
        outbuf.resize(size);
        base = (uint8_t *)&outbuf[0];
        out = base + pos;
        out_end = base + size;
        return true;
    };
    // per-file tables of a dynamic block (8 KB + 26 KB: thread-local, not on the stack of a reader thread)
    static thread_local uint32_t dyn_lit[LIT_CAP], dyn_dist[DIST_CAP];

    const uint8_t *q = in;
    for (;;) {  // members
        // ---- header (RFC 1952 2.3) ----
        if (in_end - q < 18 || q[0] != 0x1f || q[1] != 0x8b || q[2] != 8) return false;
        const uint8_t flg = q[3];
        if (flg & 0xe0) return false;  // reserved bits
        q += 10;
        if (flg & 4) {  // FEXTRA
            if (in_end - q < 2) return false;
            const size_t xlen = (size_t)q[0] | ((size_t)q[1] << 8);
            q += 2;
            if ((size_t)(in_end - q) < xlen) return false;
            q += xlen;
        }
        for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
            if (flg & f) {
                const uint8_t *z = (const uint8_t *)memchr(q, 0, (size_t)(in_end - q));
                if (!z) return false;
                q = z + 1;
            }
        if (flg & 2) {  // FHCRC
            if (in_end - q < 2) return false;
            q += 2;
        }
        const size_t member_off = (size_t)(out - base);  // (an offset: the buffer may move when it grows)
        Bits b;
        b.p = q;
        b.lim = in_end + 8;
        // ---- blocks (RFC 1951 3.2.3) ----
        for (;;) {
            if (b.p > b.lim) return false;
            b.refill();
            const uint32_t final = b.take(1), type = b.take(2);
            const uint32_t *LT, *DT;
            if (type == 0) {
                const uint8_t *s = b.byte_align();
                if (in_end - s < 4) return false;
                const size_t L = (size_t)s[0] | ((size_t)s[1] << 8), NL = (size_t)s[2] | ((size_t)s[3] << 8);
                if ((L ^ NL) != 0xffff) return false;
                s += 4;
                if ((size_t)(in_end - s) < L) return false;
                if ((size_t)(out_end - out) < L + SLACK && !grow(L + SLACK)) return false;
                memcpy(out, s, L);
                out += L;
                b.p = s + L;
                if (final) break;
                continue;
            } else if (type == 1) {
                LT = fixed.lit;
                DT = fixed.dist;
            } else if (type == 2) {
                if (!read_dynamic_header(b, dyn_lit, dyn_dist)) return false;
                add_literal_pairs(dyn_lit);
                LT = dyn_lit;
                DT = dyn_dist;
            } else {
                return false;
            }
            // ---- symbols ----
            // `e` is always the root entry of the code at the bottom of the bit buffer, looked up one step ahead: the load's latency
            // hides behind the previous symbol's stores or copy
            constexpr uint32_t LMASK = (1u << LIT_BITS) - 1;
            b.refill();
            uint32_t e = LT[b.buf & LMASK];
            for (;;) {
                if ((size_t)(out_end - out) < SLACK && !grow(SLACK)) return false;
                if (b.p > b.lim) return false;
                // here: 56+ bits in the buffer
                if ((e & KIND_MASK) == (K_SUB << 5)) {
                    b.buf >>= LIT_BITS;
                    b.cnt -= LIT_BITS;
                    e = LT[(e >> 16) + (b.buf & ((1u << e_extra(e)) - 1))];
                }
                const uint64_t saved = b.buf;
                b.buf >>= e_total(e);
                b.cnt -= (int)e_total(e);
                const uint32_t kind = e_kind(e);
                if (kind <= K_LIT2) {
                    // one or two literals (two bytes are always stored: the slack covers the extra one), then up to two more
                    // look-ups out of the bits at hand (41+, then 30+) as long as they are literals straight from the root table
                    memcpy(out, (const uint8_t *)&e + 2, 2);
                    out += 1 + kind;
                    e = LT[b.buf & LMASK];
                    if ((e & KIND_MASK) <= (K_LIT2 << 5)) {
                        b.buf >>= e_total(e);
                        b.cnt -= (int)e_total(e);
                        memcpy(out, (const uint8_t *)&e + 2, 2);
                        out += 1 + e_kind(e);
                        e = LT[b.buf & LMASK];
                        if ((e & KIND_MASK) <= (K_LIT2 << 5)) {
                            b.buf >>= e_total(e);
                            b.cnt -= (int)e_total(e);
                            memcpy(out, (const uint8_t *)&e + 2, 2);
                            out += 1 + e_kind(e);
                            b.refill();
                            e = LT[b.buf & LMASK];
                            continue;
                        }
                    }
                    b.refill();  // (`e` stays valid: a refill only adds bits above the ones it was looked up with)
                    continue;
                }
                if (kind == K_EOB) break;
                if (kind != K_BASE) return false;
                // a match: at most 15 + 5 bits are gone, 36+ left for the distance (15 + 13).  Code and extra bits left the buffer in
                // one shift; the extra bits come out of the copy
                const size_t mlen = (e >> 16) + (size_t)((saved >> e_cbits(e)) & ((1u << e_extra(e)) - 1));
                uint32_t d = DT[b.buf & ((1u << DIST_BITS) - 1)];
                if ((d & KIND_MASK) == (K_SUB << 5)) {
                    b.buf >>= DIST_BITS;
                    b.cnt -= DIST_BITS;
                    d = DT[(d >> 16) + (b.buf & ((1u << e_extra(d)) - 1))];
                }
                const uint64_t saved_d = b.buf;
                b.buf >>= e_total(d);
                b.cnt -= (int)e_total(d);
                if (e_kind(d) != K_BASE) return false;
                const size_t dist = (d >> 16) + (size_t)((saved_d >> e_cbits(d)) & ((1u << e_extra(d)) - 1));
                if (dist > (size_t)(out - base) - member_off) return false;
                b.refill();
                e = LT[b.buf & LMASK];
                const uint8_t *s = out - dist;
                if (dist >= 8) {
                    // 8 bytes at a time (up to 7 beyond the match: the slack covers them, later output overwrites them)
                    // (alignment text: most matches are shorter than 16 -- two unconditional chunks, in this order also right
                    // for 8 <= dist < 16, and a loop the branch predictor sees rarely taken)
                    memcpy(out, s, 8);
                    memcpy(out + 8, s + 8, 8);
                    for (size_t k = 16; k < mlen; k += 8) memcpy(out + k, s + k, 8);
                } else if (dist == 1) {
                    memset(out, *s, mlen);
                } else {
                    for (size_t k = 0; k < mlen; ++k) out[k] = s[k];
                }
                out += mlen;
            }
            if (final) break;
        }
        // ---- trailer: CRC-32 and ISIZE of this member ----
        const uint8_t *t = b.byte_align();
        if (t > in_end || in_end - t < 8) return false;
        const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t want_len = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        const size_t mlen = (size_t)(out - base) - member_off;
        if ((uint32_t)mlen != want_len) return false;
        if (gdca_crc32(0, base + member_off, mlen) != want_crc) return false;
        q = t + 8;
        if (q == in_end) break;
    }
    *len_out = (size_t)(out - base);
    return true;
}

// =====================================================================================================================
// One gzip member on several threads
// =====================================================================================================================
// A DEFLATE stream has no index, and a block can only be decoded knowing the 32 KB before it -- so a single .gz file is a serial
// job (43 ms for the 25 MB of config C's alignment at 600 MB/s: more than the GPU needs for the whole hot path), unless one
// speculates (the scheme of pugz, Kerbiriou & Chikhi 2019, restated for this reader):
//   1. split the compressed bytes into T pieces; in every piece but the first SEARCH the first position where a complete, valid
//      header of a non-final dynamic block parses (precode, both code-length sets: random bits pass with negligible probability);
//   2. decode every piece from its block start to the next piece's, all at once, NOT knowing the window: output symbols are 16 bits
//      wide, 0..255 a byte, 256 + k "the byte at position k of the 32 KB before this piece" (the output buffer starts with those
//      32 768 placeholders, so that a match is a plain copy whatever it points at).  Piece t must end EXACTLY at the block start
//      piece t + 1 was decoded from -- that is what validates the search;
//   3. resolve: piece 0 is bytes already; the last 32 KB of every piece are resolved one piece after the other (each needs only
//      the previous tail), then all bodies at once;
//   4. CRC-32 and ISIZE of the member over the assembled text, as always.
// Anything that does not fit -- no header found, a piece that does not land on the next start, a second member, a checksum mismatch
// -- returns false and the caller decodes the file the ordinary way.

namespace {

inline void seek_bit(Bits &b, const uint8_t *in, const uint8_t *in_end, size_t bit)
{
    b.p = in + (bit >> 3);
    b.lim = in_end + 8;
    b.buf = 0;
    b.cnt = 0;
    b.refill();
    const int d = (int)(bit & 7);
    b.buf >>= d;
    b.cnt -= d;
}

inline size_t tell_bit(const Bits &b, const uint8_t *in)
{
    return (size_t)(b.p - in) * 8 - (size_t)b.cnt;
}

// first bit position in [from, limit) where the header of a non-final dynamic block parses completely; SIZE_MAX: none
size_t find_block_start(const uint8_t *in, const uint8_t *in_end, size_t from, size_t limit, uint32_t *lit, uint32_t *dist)
{
    for (size_t bit = from; bit < limit; ++bit) {
        const uint64_t w = load64(in + (bit >> 3)) >> (bit & 7);  // 57+ bits from `bit` on
        if ((w & 7) != 4) continue;                                 // BFINAL = 0, BTYPE = 10b
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;  // HLIT, HDIST
        // the code-length code must be complete (Kraft sum = 1): 13 of its (up to 19) lengths are in this word already
        const int hclen = (int)((w >> 13) & 15) + 4;
        int kraft = 0;
        const int here = hclen < 13 ? hclen : 13;
        for (int i = 0; i < here; ++i) {
            const int l = (int)((w >> (17 + 3 * i)) & 7);
            if (l) kraft += 128 >> l;
        }
        if (kraft > 128 || (hclen <= 13 && kraft != 128)) continue;
        Bits b;
        seek_bit(b, in, in_end, bit);
        b.take(3);
        if (read_dynamic_header(b, lit, dist)) return bit;
    }
    return SIZE_MAX;
}

// decode blocks from `start_bit` until a block would start at `stop_bit` (SIZE_MAX: until the final block has ended); Out = uint8_t
// (no window before the piece: the beginning of the member) or uint16_t (unknown window: `out` begins with 32 768 placeholders)
template <class Out>
bool decode_piece(const uint8_t *in, const uint8_t *in_end, size_t start_bit, size_t stop_bit, std::vector<Out> &out, size_t prefix, size_t max_elems,
                  size_t *produced, const uint8_t **after_final)
{
    static const FixedTables fixed;
    if (!fixed.ok) return false;
    static thread_local uint32_t dyn_lit[LIT_CAP], dyn_dist[DIST_CAP];
    constexpr size_t SLACK = 320;
    constexpr uint32_t LMASK = (1u << LIT_BITS) - 1;
    size_t pos = prefix;  // next element of `out`
    auto room = [&](size_t want) {
        if (out.size() - pos >= want) return true;
        size_t size = out.size();
        while (size - pos < want) size *= 2;
        if (size > max_elems) return false;  // (DEFLATE cannot expand by more than 1032 : 1: a piece that wants more is not what it seems)
        out.resize(size);
        return true;
    };
    Bits b;
    seek_bit(b, in, in_end, start_bit);
    *after_final = nullptr;
    for (;;) {
        const size_t at = tell_bit(b, in);
        if (at == stop_bit) break;
        if (at > stop_bit || b.p > b.lim) return false;
        b.refill();
        const uint32_t final = b.take(1), type = b.take(2);
        const uint32_t *LT, *DT;
        if (type == 0) {
            const uint8_t *s = b.byte_align();
            if (in_end - s < 4) return false;
            const size_t L = (size_t)s[0] | ((size_t)s[1] << 8), NL = (size_t)s[2] | ((size_t)s[3] << 8);
            if ((L ^ NL) != 0xffff) return false;
            s += 4;
            if ((size_t)(in_end - s) < L || !room(L + SLACK)) return false;
            for (size_t k = 0; k < L; ++k) out[pos + k] = (Out)s[k];
            pos += L;
            seek_bit(b, in, in_end, (size_t)(s + L - in) * 8);
        } else {
            if (type == 1) {
                LT = fixed.lit;
                DT = fixed.dist;
            } else if (type == 2) {
                if (!read_dynamic_header(b, dyn_lit, dyn_dist)) return false;
                add_literal_pairs(dyn_lit);
                LT = dyn_lit;
                DT = dyn_dist;
            } else {
                return false;
            }
            for (;;) {
                if (!room(SLACK) || b.p > b.lim) return false;
                Out *o = out.data() + pos;
                b.refill();
                uint32_t e = LT[b.buf & LMASK];
                if ((e & KIND_MASK) == (K_SUB << 5)) {
                    b.buf >>= LIT_BITS;
                    b.cnt -= LIT_BITS;
                    e = LT[(e >> 16) + (b.buf & ((1u << e_extra(e)) - 1))];
                }
                const uint64_t saved = b.buf;
                b.buf >>= e_total(e);
                b.cnt -= (int)e_total(e);
                const uint32_t kind = e_kind(e);
                if (kind <= K_LIT2) {
                    o[0] = (Out)((e >> 16) & 0xff);
                    o[1] = (Out)(e >> 24);
                    pos += 1 + kind;
                    // one more look-up out of the bits at hand (41+) where it is a literal again
                    e = LT[b.buf & LMASK];
                    if ((e & KIND_MASK) <= (K_LIT2 << 5)) {
                        b.buf >>= e_total(e);
                        b.cnt -= (int)e_total(e);
                        o = out.data() + pos;
                        o[0] = (Out)((e >> 16) & 0xff);
                        o[1] = (Out)(e >> 24);
                        pos += 1 + e_kind(e);
                    }
                    continue;
                }
                if (kind == K_EOB) break;
                if (kind != K_BASE) return false;
                const size_t mlen = (e >> 16) + (size_t)((saved >> e_cbits(e)) & ((1u << e_extra(e)) - 1));
                uint32_t d = DT[b.buf & ((1u << DIST_BITS) - 1)];
                if ((d & KIND_MASK) == (K_SUB << 5)) {
                    b.buf >>= DIST_BITS;
                    b.cnt -= DIST_BITS;
                    d = DT[(d >> 16) + (b.buf & ((1u << e_extra(d)) - 1))];
                }
                const uint64_t saved_d = b.buf;
                b.buf >>= e_total(d);
                b.cnt -= (int)e_total(d);
                if (e_kind(d) != K_BASE) return false;
                const size_t dist = (d >> 16) + (size_t)((saved_d >> e_cbits(d)) & ((1u << e_extra(d)) - 1));
                if (dist > pos) return false;  // (before the placeholders / before the member's first byte)
                const Out *src = o - dist;
                if (dist >= 8) {
                    // 8 elements at a time, two chunks unconditionally (most matches of alignment text are shorter than 16)
                    memcpy(o, src, 8 * sizeof(Out));
                    memcpy(o + 8, src + 8, 8 * sizeof(Out));
                    for (size_t k = 16; k < mlen; k += 8) memcpy(o + k, src + k, 8 * sizeof(Out));
                } else {
                    for (size_t k = 0; k < mlen; ++k) o[k] = src[k];
                }
                pos += mlen;
            }
        }
        if (final) {
            *after_final = b.byte_align();
            break;
        }
    }
    *produced = pos - prefix;
    // a piece that was to stop at a block start must not have met the final block, and the other way round
    return (stop_bit == SIZE_MAX) == (*after_final != nullptr);
}

}  // namespace

namespace {
// fn(t) for t = first .. T-1 on threads of their own; false if a thread could not be started or one of them threw (out of memory
// while a piece's buffer grew): nothing escapes into the caller's thread, which then takes the serial path
template <class F>
bool on_threads(int first, int T, F fn)
{
    std::vector<std::thread> th;
    std::vector<char> failed((size_t)T, 0);
    bool started = true;
    try {
        for (int t = first; t < T; ++t)
            th.emplace_back([&, t] {
                try {
                    fn(t);
                } catch (...) {
                    failed[(size_t)t] = 1;
                }
            });
    } catch (...) {
        started = false;
    }
    for (auto &x : th) x.join();
    if (!started) return false;
    for (char f : failed)
        if (f) return false;
    return true;
}
}  // namespace

bool gdca_gunzip_parallel(const uint8_t *in, size_t n, std::string &outbuf, size_t *len_out, size_t hint, int threads)
try {
    constexpr size_t WIN = 32768, MIN_PIECE = (size_t)512 << 10;
    const uint8_t *const in_end = in + n;
    // ---- header of the (single) member ----
    const uint8_t *q = in;
    // (files beyond 1 GiB compressed stay with the serial decoder: the pieces' 16-bit images would take twice the text's size on top of it)
    if (n < 18 + 2 * MIN_PIECE || n > ((size_t)1 << 30) || q[0] != 0x1f || q[1] != 0x8b || q[2] != 8) return false;
    const uint8_t flg = q[3];
    if (flg & 0xe0) return false;
    q += 10;
    if (flg & 4) {
        if (in_end - q < 2) return false;
        const size_t xlen = (size_t)q[0] | ((size_t)q[1] << 8);
        q += 2;
        if ((size_t)(in_end - q) < xlen) return false;
        q += xlen;
    }
    for (int f = 8; f <= 16; f <<= 1)
        if (flg & f) {
            const uint8_t *z = (const uint8_t *)memchr(q, 0, (size_t)(in_end - q));
            if (!z) return false;
            q = z + 1;
        }
    if (flg & 2) {
        if (in_end - q < 2) return false;
        q += 2;
    }
    if (in_end - q < (ptrdiff_t)(8 + 2 * MIN_PIECE)) return false;
    const size_t first_bit = (size_t)(q - in) * 8, last_bit = (size_t)(in_end - 8 - in) * 8;
    int T = (int)std::min<size_t>((size_t)std::max(threads, 1), (last_bit - first_bit) / 8 / MIN_PIECE);
    if (T < 2) return false;
    static const bool trace = getenv("GDCA_INFLATE_TRACE") != nullptr;
    auto tick = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = trace ? tick() : 0.0;
    // ---- 1. block starts ----
    std::vector<size_t> start((size_t)T + 1, SIZE_MAX);
    start[0] = first_bit;
    start[(size_t)T] = SIZE_MAX;
    if (!on_threads(1, T, [&](int t) {
            static thread_local uint32_t lit[LIT_CAP], dist[DIST_CAP];
            const size_t from = first_bit + (last_bit - first_bit) / (size_t)T * (size_t)t;
            const size_t limit = first_bit + (last_bit - first_bit) / (size_t)T * (size_t)(t + 1);
            start[(size_t)t] = find_block_start(in, in_end, from, limit, lit, dist);
        }))
        return false;
    for (int t = 1; t < T; ++t)
        if (start[(size_t)t] == SIZE_MAX) return false;
    const double t_1 = trace ? tick() : 0.0;
    // ---- 2. every piece, window unknown ----
    std::vector<uint8_t> first;
    std::vector<std::vector<uint16_t>> piece((size_t)T);
    std::vector<size_t> produced((size_t)T, 0);
    std::vector<char> ok((size_t)T, 0);
    const uint8_t *after_final = nullptr;
    // (the size hint is the trailer's ISIZE, untrusted: alignment text deflates 2-10 x; a file that claims more than 16 x is left to the
    // serial decoder instead of sizing sixteen 16-bit buffers after it)
    if (hint > 16 * n) return false;
    const size_t guess = std::max<size_t>((hint ? hint : 4 * n) / (size_t)T * 5 / 4, (size_t)1 << 20);
    if (!on_threads(0, T, [&](int t) {
                const uint8_t *af = nullptr;
                const size_t piece_bytes = ((t + 1 < T ? start[(size_t)t + 1] : last_bit) - start[(size_t)t]) / 8 + 16;
                const size_t cap = 2 * (WIN + 1040 * piece_bytes + ((size_t)1 << 20));   // (x 2: the buffer grows by doubling)
                if (t == 0) {
                    first.resize(std::min(guess, cap));
                    ok[0] = decode_piece<uint8_t>(in, in_end, start[0], start[1], first, 0, cap, &produced[0], &af);
                } else {
                    std::vector<uint16_t> &v = piece[(size_t)t];
                    v.resize(WIN + std::min(guess, cap));
                    for (size_t k = 0; k < WIN; ++k) v[k] = (uint16_t)(256 + k);
                    ok[(size_t)t] = decode_piece<uint16_t>(in, in_end, start[(size_t)t], start[(size_t)t + 1], v, WIN, cap, &produced[(size_t)t], &af);
                }
                if (t == T - 1) after_final = af;
        }))
        return false;
    const double t_2 = trace ? tick() : 0.0;
    size_t total = 0;
    for (int t = 0; t < T; ++t) {
        if (!ok[(size_t)t] || produced[(size_t)t] < WIN) return false;
        total += produced[(size_t)t];
    }
    // the member must end here: trailer, end of file
    if (!after_final || after_final != in_end - 8) return false;
    const uint8_t *tr = after_final;
    const uint32_t want_crc = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
    const uint32_t want_len = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
    if ((uint32_t)total != want_len) return false;
    // ---- 3. resolve ----
    if (outbuf.size() < total + 64) outbuf.resize(total + 64);
    uint8_t *base = (uint8_t *)&outbuf[0];
    std::vector<size_t> off((size_t)T + 1, 0);
    for (int t = 0; t < T; ++t) off[(size_t)t + 1] = off[(size_t)t] + produced[(size_t)t];
    memcpy(base, first.data(), produced[0]);
    auto resolve = [&](int t, size_t a, size_t b) {  // elements [a, b) of piece t
        const uint16_t *src = piece[(size_t)t].data() + WIN;
        const uint8_t *win = base + off[(size_t)t] - WIN;
        uint8_t *dst = base + off[(size_t)t];
        for (size_t k = a; k < b; ++k) {
            const uint16_t v = src[k];
            dst[k] = v < 256 ? (uint8_t)v : win[v - 256];
        }
    };
    for (int t = 1; t < T; ++t) resolve(t, produced[(size_t)t] - WIN, produced[(size_t)t]);  // the tails, one after the other
    // ---- 4. checksum: every piece's CRC-32 by the thread that resolved it, combined in order ----
    std::vector<uint32_t> crc((size_t)T, 0);
    if (!on_threads(0, T, [&](int t) {
            if (t > 0) resolve(t, 0, produced[(size_t)t] - WIN);
            crc[(size_t)t] = gdca_crc32(0, base + off[(size_t)t], produced[(size_t)t]);
        }))
        return false;
    const double t_3 = trace ? tick() : 0.0;
    uint32_t whole = crc[0];
    for (int t = 1; t < T; ++t) whole = gdca_crc32_combine(whole, crc[(size_t)t], produced[(size_t)t]);
    if (whole != want_crc) return false;
    if (trace)
        fprintf(stderr, "inflate-trace %d threads: block starts %.2f ms, pieces %.2f ms, resolve + crc %.2f ms, combine %.2f ms\n", T, (t_1 - t_0) * 1e3, (t_2 - t_1) * 1e3,
                (t_3 - t_2) * 1e3, (tick() - t_3) * 1e3);
    *len_out = total;
    return true;
} catch (...) {
    return false;  // (out of memory sizing a buffer: the serial decoder needs a fraction of it)
}
