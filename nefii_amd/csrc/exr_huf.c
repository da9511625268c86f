/* Host helper of nefii_amd/utils/exr.py: the symbol loop of the PIZ Huffman decoder (ImfHuf.cpp hufDecode / getCode:
 * MSB-first bit stream, codes up to 58 bits, symbol `rlc` + an 8-bit count repeats the previous symbol).  The Python
 * reader builds the canonical code table and calls this for the ~1 M symbols of a chunk; without the library it runs
 * the same loop in Python (~50x slower).  Plain C, built by nefii_amd/build.py with the system compiler. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

/* lengths[n_sym], codes[n_sym]: code length (0 = unused) and canonical code of every symbol.
 * Returns the number of symbols written (== n_out on success), or -1 on a malformed stream. */
long nefii_exr_huf_decode(const uint8_t *data, size_t n_bytes, long n_bits, const int64_t *lengths,
                          const int64_t *codes, long n_sym, long rlc, uint16_t *out, long n_out) {
    enum { DEC = 14 };
    static const long TSIZE = 1L << DEC;
    int32_t sym_t[1 << DEC];
    uint8_t len_t[1 << DEC];
    for (long i = 0; i < TSIZE; ++i) sym_t[i] = 0, len_t[i] = 0;
    int max_len = 0;
    /* codes of one length are consecutive integers handed out in symbol order (canonical code): per length the first
     * code, the count and where that length's symbols start in `by_len` */
    int64_t first[64];
    long count[64], start[64];
    for (int l = 0; l < 64; ++l) first[l] = 0, count[l] = 0, start[l] = 0;
    for (long s = 0; s < n_sym; ++s) {
        const int l = (int)lengths[s];
        if (l <= 0 || l >= 64) continue;
        if (count[l] == 0 || codes[s] < first[l]) first[l] = codes[s];
        ++count[l];
        if (l > max_len) max_len = l;
    }
    long total = 0;
    for (int l = 1; l < 64; ++l) start[l] = total, total += count[l];
    int32_t *by_len = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    if (!by_len) return -1;
    for (long s = 0; s < n_sym; ++s) {
        const int l = (int)lengths[s];
        if (l <= 0 || l >= 64) continue;
        /* an over-subscribed length table yields codes that do not fit their length (ImfHuf.cpp hufBuildDecTable rejects
         * it with `if (c >> l)`): such a code would index past the decoding tables */
        if (codes[s] < 0 || (codes[s] >> l) != 0 || codes[s] - first[l] >= count[l]) { free(by_len); return -1; }
        by_len[start[l] + (long)(codes[s] - first[l])] = (int32_t)s;
        if (l <= DEC) {
            const long base = (long)(codes[s] << (DEC - l)), span = 1L << (DEC - l);
            for (long k = 0; k < span; ++k) sym_t[base + k] = (int32_t)s, len_t[base + k] = (uint8_t)l;
        }
    }
    long bp = 0, o = 0;
    while (bp < n_bits && o < n_out) {
        /* the 64 bits starting at bit bp */
        const size_t byte = (size_t)(bp >> 3);
        const unsigned sh = (unsigned)(bp & 7);
        uint64_t w = 0;
        for (int k = 0; k < 8; ++k) w = (w << 8) | (byte + k < n_bytes ? data[byte + k] : 0);
        if (sh) w = (w << sh) | ((uint64_t)(byte + 8 < n_bytes ? data[byte + 8] : 0) >> (8 - sh));
        const long pre = (long)(w >> (64 - DEC));
        int l = len_t[pre];
        long s = -1;
        if (l) {
            s = sym_t[pre];
        } else {
            for (l = DEC + 1; l <= max_len; ++l) {
                const int64_t c = (int64_t)(w >> (64 - l));
                if (count[l] && c >= first[l] && c < first[l] + count[l]) {
                    s = by_len[start[l] + (long)(c - first[l])];
                    break;
                }
            }
            if (s < 0) { free(by_len); return -1; }
        }
        bp += l;
        if (s == rlc) {
            const size_t b2 = (size_t)(bp >> 3);
            const unsigned v = ((unsigned)(b2 < n_bytes ? data[b2] : 0) << 8) | (unsigned)(b2 + 1 < n_bytes ? data[b2 + 1] : 0);
            const long cnt = (v >> (8 - (bp & 7))) & 0xff;
            bp += 8;
            if (o == 0 || o + cnt > n_out) { free(by_len); return -1; }
            for (long k = 0; k < cnt; ++k) out[o + k] = out[o - 1];
            o += cnt;
        } else {
            out[o++] = (uint16_t)s;
        }
    }
    free(by_len);
    return o;
}
