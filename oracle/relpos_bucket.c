/* CPU ORACLE (test infrastructure): C restatement of the T5 relative-position bucket function.
 *
 * Follows HF models/t5/modeling_t5.py:216-262 (`T5Attention._relative_position_bucket`), the
 * third-party code the reference executes for clip-flant5 (transformers>=4.52.0,
 * /root/reference/pyproject.toml:26).  Integer in, integer out; the logarithmic branch uses the
 * same fp32 operation order as torch: logf(float(n)/max_exact) / float(log(max_distance/max_exact))
 * * (nb - max_exact), truncated toward zero.
 *
 * Pinned by tests/golden/relpos_buckets.npz (produced from the HF function by oracle/make_golden.py).
 * Built by __graft_entry__.build() into oracle/_build/librelpos_oracle.so; used only by tests.
 */
#include <math.h>
#include <stdint.h>

int32_t t5_relpos_bucket(int32_t relative_position, int32_t bidirectional, int32_t num_buckets,
                         int32_t max_distance)
{
    int32_t bucket = 0;
    int32_t nb = num_buckets;
    int32_t rp = relative_position;
    if (bidirectional) {
        nb /= 2;
        if (rp > 0) bucket += nb;
        if (rp < 0) rp = -rp;
    } else {
        rp = rp < 0 ? -rp : 0;
    }
    int32_t max_exact = nb / 2;
    if (rp < max_exact) return bucket + rp;
    float q = (float)rp / (float)max_exact;
    float v = logf(q) / (float)log((double)max_distance / (double)max_exact);
    v = v * (float)(nb - max_exact);
    int32_t large = max_exact + (int32_t)v;
    if (large > nb - 1) large = nb - 1;
    return bucket + large;
}

void t5_relpos_bucket_many(const int32_t* rel, int32_t n, int32_t bidirectional, int32_t num_buckets,
                           int32_t max_distance, int32_t* out)
{
    for (int32_t i = 0; i < n; ++i)
        out[i] = t5_relpos_bucket(rel[i], bidirectional, num_buckets, max_distance);
}
