/* Accelerator for breakfast_amd/synth.py (the seeded synthetic workloads of bench.py and the tests; NOT part of the
 * product path): the two inner loops that dominate the generation of 10M-100M reads, in C with OpenMP.  Both produce
 * exactly the bytes of the numpy code they stand in for (tests/test_host.py compares them). */
#include <stdint.h>

/* synth.make_reads, build_chunk: read r, position k (ko = k, or L-1-k when the read is presented flipped) comes from
 * the 5' side (ko < b) at cat[base1 + dir1 * ko] or from the 3' side at cat[base2 + dir2 * ko], complemented when the
 * side lies on the minus strand XOR the read is flipped. */
void bf_synth_fill(uint8_t* seq, const uint8_t* cat, const int64_t* base1, const int64_t* base2, const int64_t* dir1,
                   const int64_t* dir2, const int64_t* b, const uint8_t* flip, const uint8_t* s1, const uint8_t* s2,
                   int64_t m, int64_t L, const uint8_t* comp) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; r++) {
        uint8_t* out = seq + r * L;
        const int f = flip[r] != 0;
        const int c1 = (s1[r] == 0) ^ f, c2 = (s2[r] == 0) ^ f;
        const int64_t bb = b[r], b1 = base1[r], b2 = base2[r], d1 = dir1[r], d2 = dir2[r];
        for (int64_t k = 0; k < L; k++) {
            const int64_t ko = f ? L - 1 - k : k;
            const int left = ko < bb;
            const uint8_t x = cat[left ? b1 + d1 * ko : b2 + d2 * ko];
            out[k] = (left ? c1 : c2) ? comp[x] : x;
        }
    }
}

/* engine.pack_batch for reads of one length: ASCII -> BAM 4-bit codes, two bases a byte, high nibble first, every
 * read on a byte boundary.  Returns the number of bytes outside "=ACMGRSVTWYHKDBN" (then the output is meaningless). */
int64_t bf_synth_pack_fixed(uint8_t* out, const uint8_t* seq, int64_t n, int64_t L) {
    static const char codes[] = "=ACMGRSVTWYHKDBN";
    uint8_t lut[256];
    for (int i = 0; i < 256; i++) lut[i] = 255;
    for (int i = 0; i < 16; i++) lut[(uint8_t)codes[i]] = (uint8_t)i;
    const int64_t nb = (L + 1) / 2;
    int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int64_t r = 0; r < n; r++) {
        const uint8_t* s = seq + r * L;
        uint8_t* o = out + r * nb;
        for (int64_t k = 0; k + 1 < L; k += 2) {
            const uint8_t a = lut[s[k]], c = lut[s[k + 1]];
            bad += (a == 255) + (c == 255);
            o[k >> 1] = (uint8_t)((a << 4) | (c & 15));
        }
        if (L & 1) { const uint8_t a = lut[s[L - 1]]; bad += a == 255; o[nb - 1] = (uint8_t)(a << 4); }
    }
    return bad;
}

/* engine.pack2_batch for reads of one length: BAM 4-bit codes -> 2-bit codes (A C G T = 0 1 2 3, base k in bits 2(k & 3)..
 * of byte k >> 2, every read on a byte).  Returns the number of bases that are none of the four (the caller then takes
 * the general path, which lists them); the output is complete only when that is zero. */
int64_t bf_synth_pack2_fixed(uint8_t* out, const uint8_t* nib, int64_t n, int64_t L) {
    static const uint8_t two[16] = {255, 0, 1, 255, 2, 255, 255, 255, 3, 255, 255, 255, 255, 255, 255, 255};
    const int64_t nb4 = (L + 1) / 2, nb2 = (L + 3) / 4;
    int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int64_t r = 0; r < n; r++) {
        const uint8_t* s = nib + r * nb4;
        uint8_t* o = out + r * nb2;
        for (int64_t b = 0; b < nb2; b++) {
            uint8_t x = 0;
            for (int64_t j = 0; j < 4 && 4 * b + j < L; j++) {
                const int64_t k = 4 * b + j;
                const uint8_t c = two[(k & 1) ? (s[k >> 1] & 15) : (s[k >> 1] >> 4)];
                bad += c == 255;
                x |= (uint8_t)((c & 3) << (2 * j));
            }
            o[b] = x;
        }
    }
    return bad;
}
