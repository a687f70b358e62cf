// A gzip (RFC 1952) / DEFLATE (RFC 1951) decoder for the FASTA reader's .gz path (DCAUtils.read_fasta_alignment reads .gz files
// transparently; the reference's own test data are .fasta.gz).  zlib's inflate() decodes one symbol per loop trip through a byte-wise
// bit reader: 350 MB/s of output per thread on the GPU box's host for alignment text (ratio ~2: mostly literals), which made the
// eight-GPU batch driver's gzip feed (370 families/s on 16 CPUs) slower than the GPUs (8 x 68).  This one keeps 56+ bits in a
// register, looks codes up in an 11-bit table, and emits up to three literals per refill.
//
// It decodes into ONE growing buffer and verifies CRC-32 and ISIZE of every member.  It is a fast path only: whatever it does
// not like -- a malformed or truncated stream, a checksum mismatch, a code it considers invalid -- makes it return false, and the
// caller decodes the file again with zlib, whose verdict (and error behaviour) stands.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

// in[0 .. n): the whole .gz file (one or more members); the caller guarantees GDCA_INFLATE_PAD readable bytes after in[n-1].
// out: resized as needed (never shrunk), *len = bytes produced.  hint: expected size (0 = unknown).
#define GDCA_INFLATE_PAD 16
bool gdca_gunzip_fast(const uint8_t *in, size_t n, std::string &out, size_t *len, size_t hint);
// One single-member file on `threads` threads (speculative block starts, 16-bit symbols for the unknown windows, resolved afterwards:
// see gdca_inflate.cpp); same contract as gdca_gunzip_fast, false also for small files, several members, or a failed speculation.
bool gdca_gunzip_parallel(const uint8_t *in, size_t n, std::string &out, size_t *len, size_t hint, int threads);
// CRC-32 (IEEE 802.3, as in gzip trailers): carry-less-multiply folding where the CPU has PCLMULQDQ, slicing-by-16 tables otherwise
uint32_t gdca_crc32(uint32_t crc, const uint8_t *p, size_t n);
// CRC-32 of the concatenation A || B from crc(A), crc(B) and the length of B
uint32_t gdca_crc32_combine(uint32_t crc_a, uint32_t crc_b, size_t len_b);
