/* C oracle for the CSPN affinity-propagation hot path.  TEST INFRASTRUCTURE ONLY:
 * linked/loaded solely by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Plain scalar C restatement of the reference's arithmetic (see cspn_oracle_impl.inc for the
 * file:line map); pinned against the tests/golden fixtures by tests/test_oracle_golden.py.
 * Build: make -C oracle   ->  oracle/libcspn_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* o_k = (dy,dx) of reference plane k: CSPN_new.py:43-67 pad table after the crop at :88 */
static const int ORC_OFF[8][2] = {{+1, +1}, {+1, 0}, {+1, -1}, {0, +1}, {0, -1}, {-1, +1}, {-1, 0}, {-1, -1}};

#define REAL float
#define SUF f32
#include "cspn_oracle_impl.inc"
#undef REAL
#undef SUF

#define REAL double
#define SUF f64
#include "cspn_oracle_impl.inc"
#undef REAL
#undef SUF

/* ---- integer-hash input generator (mirrors oracle/cspn_oracle.py hash_*) ---- */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint32_t hash_u24(uint64_t seed, uint64_t tid, uint64_t stream, uint64_t idx) {
    const uint64_t key = seed * 0x100000001B3ull + tid * 0x9E3779B1ull + stream * 0x85EBCA77ull;
    return (uint32_t)(splitmix64(splitmix64(key) ^ idx) >> 40);
}

void orc_hash_uniform_f32(uint64_t seed, uint64_t tid, size_t n, double lo, double hi, float* out) {
    for (size_t i = 0; i < n; ++i)
        out[i] = (float)(lo + (hi - lo) * ((double)hash_u24(seed, tid, 0, i) / 16777216.0));
}

void orc_hash_normal_f32(uint64_t seed, uint64_t tid, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) {
        double s = 0;
        for (int st = 1; st <= 4; ++st) s += (double)hash_u24(seed, tid, st, i);
        out[i] = (float)((s / 16777216.0 - 2.0) * 1.7320508075688772);
    }
}

void orc_hash_sparse_f32(uint64_t seed, uint64_t tid, size_t n, double keep_prob, const float* depth, float* out) {
    for (size_t i = 0; i < n; ++i)
        out[i] = ((double)hash_u24(seed, tid, 0, i) / 16777216.0 < keep_prob) ? depth[i] : 0.0f;
}
