/* Synthetic input generator (not oracle, not product): "wide Fibonacci" execution trace for
 * tools/synth_pil.py -- W independent pairs (a_k, b_k) per row, a' = b, b' = a + b (mod p),
 * row-major [N][2W] little-endian u64, plus splitmix64 columns for the NTT benchmark.
 * Build: gcc -O2 -shared -fPIC tools/tracegen.c -o tools/libtracegen.so */
#include <stdint.h>
#define P 0xFFFFFFFF00000001ULL
static inline uint64_t addp(uint64_t a, uint64_t b) { unsigned __int128 s = (unsigned __int128)a + b; if (s >= P) s -= P; return (uint64_t)s; }
/* first row a_k = seed + k + 1, b_k = 2k + 3: every seed is a different valid witness (sub-proof input) */
void widefib_trace_seed(unsigned nbits, unsigned W, uint64_t seed, uint64_t *out) {
    uint64_t N = (uint64_t)1 << nbits;
    for (unsigned k = 0; k < W; ++k) { out[2 * k] = (seed + k + 1) % P; out[2 * k + 1] = 2 * k + 3; }
    for (uint64_t i = 1; i < N; ++i)
        for (unsigned k = 0; k < W; ++k) {
            const uint64_t *p = out + (i - 1) * 2 * W + 2 * k;
            uint64_t *c = out + i * 2 * W + 2 * k;
            c[0] = p[1]; c[1] = addp(p[0], p[1]);
        }
}
void widefib_trace(unsigned nbits, unsigned W, uint64_t *out) { widefib_trace_seed(nbits, W, 0, out); }
