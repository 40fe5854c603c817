/* C restatement of the integer/byte work on the path (TEST INFRASTRUCTURE ONLY; never linked by the product).
 *   mu-law quantise / expand : mimikit/features/functionals.py:330-338, :361-369
 *   greedy decode            : mimikit/modules/targets.py:41-42 (argmax, first maximum wins)
 * Straight fp32 evaluation of the reference formulas with libm.  libm's log1pf and torch's vectorised log1p
 * may differ in the last bit, which can move a sample that sits within an ulp of a bin edge to the neighbouring
 * code; tests/test_oracle_c.py pins this file to the reference's golden codes exactly away from the edges and
 * to +-1 code on the edge-adjacent probes.  Build: `make -C oracle` -> oracle/_build/libmulaw_oracle.so */
#include <math.h>
#include <stdint.h>

static float signf_(float x) { return (x > 0.f) - (x < 0.f); }

void oracle_mulaw_compress(const float* x, int64_t* codes, int64_t n, int q_levels, float compression) {
  const float mu = (float)(q_levels - 1);
  const float denom = log1pf(mu * compression);
  for (int64_t i = 0; i < n; ++i) {
    const float y = signf_(x[i]) * log1pf(mu * fabsf(x[i]) * compression) / denom;
    codes[i] = (int64_t)((y + 1.f) / 2.f * mu + 0.5f); /* truncation toward zero, as tensor.to(int64) */
  }
}

void oracle_mulaw_expand(const int64_t* codes, float* x, int64_t n, int q_levels, float compression) {
  const float mu = (float)(q_levels - 1);
  const float l = log1pf(mu * compression);
  for (int64_t i = 0; i < n; ++i) {
    const float v = ((float)codes[i] / mu) * 2.f - 1.0f;
    x[i] = signf_(v) * (expf(fabsf(v) * l) - 1.0f) / (mu * compression);
  }
}

void oracle_argmax_rows(const float* logits, int64_t rows, int64_t cols, int64_t ld, int64_t* out) {
  for (int64_t r = 0; r < rows; ++r) {
    const float* p = logits + r * ld;
    int64_t best = 0;
    for (int64_t c = 1; c < cols; ++c)
      if (p[c] > p[best]) best = c;
    out[r] = best;
  }
}
