// Host-side batch sampler: the per-clip draw loop of AudioProcessor.get_data (reference
// input_data.py:457-514) as native code.  It consumes NumPy's GLOBAL legacy RNG
// stream exactly like the reference's Python loop does - the caller passes in the MT19937 state of
// np.random.get_state() and writes the advanced state back - so a seeded run draws the same clips,
// shifts and volumes as the reference, at ~100x the speed of the interpreter loop (which otherwise
// caps the pipeline near 130 k clips/s).
//   np.random.uniform(a, b)  = a + (b - a) * rk_double,  rk_double = ((r1 >> 5) * 2^26 + (r2 >> 6)) / 2^53
//   np.random.randint(lo, hi) = lo + masked rejection sampling of 32-bit draws on [0, hi - lo - 1]
//                               (no draw at all when the range is a single value)
#include <stdint.h>

#include "../../include/kws_hip.h"

void kws_set_error(const char* fmt, ...);

namespace {

struct MT {
  uint32_t* key;
  int pos;
  uint32_t next() {
    if (pos >= 624) {
      const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
      int i;
      uint32_t y;
      for (i = 0; i < 624 - 397; i++) {
        y = (key[i] & UPPER) | (key[i + 1] & LOWER);
        key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAG);
      }
      for (; i < 623; i++) {
        y = (key[i] & UPPER) | (key[i + 1] & LOWER);
        key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAG);
      }
      y = (key[623] & UPPER) | (key[0] & LOWER);
      key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAG);
      pos = 0;
    }
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  double next_double() {
    const int32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  double uniform(double lo, double hi) { return lo + (hi - lo) * next_double(); }
  // legacy RandomState.randint(lo, hi): hi exclusive
  int64_t randint(int64_t lo, int64_t hi) {
    const uint64_t rng = (uint64_t)(hi - lo - 1);
    if (rng == 0) return lo;
    if (rng == 0xFFFFFFFFull) return lo + (int64_t)next();
    uint64_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    if (rng <= 0xFFFFFFFFull) {
      uint32_t v;
      do {
        v = next() & (uint32_t)mask;
      } while (v > rng);
      return lo + (int64_t)v;
    }
    uint64_t v;
    do {
      v = (((uint64_t)next() << 32) | next()) & mask;
    } while (v > rng);
    return lo + (int64_t)v;
  }
};

}  // namespace

extern "C" int kws_sampler_draw(uint32_t* mt_key, int* mt_pos, const kws_sampler_set_t* cand,
                                const kws_sampler_set_t* pseudo, const kws_sampler_args_t* a, int32_t* out_rows,
                                int32_t* out_labels, int32_t* out_shift, int64_t* out_bg_off, float* out_bg_vol,
                                float* out_fg_vol) {
  if (!mt_key || !mt_pos || !cand || !pseudo || !a || !out_rows || !out_labels || !out_shift || !out_bg_off ||
      !out_bg_vol || !out_fg_vol) {
    kws_set_error("sampler_draw: NULL pointer");
    return KWS_E_INVALID;
  }
  if (a->count < 0 || a->offset < 0 || (a->deterministic && a->offset + a->count > cand->n) ||
      (!a->deterministic && cand->n <= 0) || (a->use_background && (a->n_bg <= 0 || !a->bg_len || !a->bg_start))) {
    kws_set_error("sampler_draw: inconsistent arguments (count=%d offset=%d n=%d)", a->count, a->offset, cand->n);
    return KWS_E_INVALID;
  }
  MT mt{mt_key, *mt_pos};
  for (int k = 0; k < a->count; ++k) {
    int32_t row, lab;
    bool sil;
    if (a->deterministic) {
      const int i = a->offset + k;
      row = cand->rows[i]; lab = cand->labels[i]; sil = cand->silence[i] != 0;
    } else if (mt.uniform(0, 1) < a->pseudo_frequency) {
      if (pseudo->n <= 0) {
        kws_set_error("sampler_draw: pseudo partition is empty");
        return KWS_E_INVALID;
      }
      const int64_t j = mt.randint(0, pseudo->n);
      row = pseudo->rows[j]; lab = pseudo->labels[j]; sil = pseudo->silence[j] != 0;
    } else {
      const int64_t j = mt.randint(0, cand->n);
      row = cand->rows[j]; lab = cand->labels[j]; sil = cand->silence[j] != 0;
    }
    int32_t shift = 0;
    if (mt.uniform(0.0, 1.0) < a->time_shift_frequency) {
      if (a->shift_lo > a->shift_hi) {
        kws_set_error("sampler_draw: low >= high (time_shift_range [%d, %d])", a->shift_lo, a->shift_hi);
        *mt_pos = mt.pos;
        return KWS_E_INVALID;
      }
      shift = (int32_t)mt.randint(a->shift_lo, (int64_t)a->shift_hi + 1);
    }
    int64_t bg_off = 0;
    double bg_vol = 0.0;
    if (a->use_background) {
      const int64_t bi = mt.randint(0, a->n_bg);
      if (a->bg_len[bi] <= (int64_t)a->desired_samples) {
        // np.random.randint(0, n <= 0) raises ValueError("low >= high") in the reference (input_data.py:485);
        // the unsigned range below would wrap and index outside the recording
        kws_set_error("sampler_draw: low >= high (background recording %lld has %lld samples, desired_samples %d)",
                      (long long)bi, (long long)a->bg_len[bi], a->desired_samples);
        *mt_pos = mt.pos;
        return KWS_E_INVALID;
      }
      const int64_t bo = mt.randint(0, a->bg_len[bi] - a->desired_samples);
      bg_off = a->bg_start[bi] + bo;
      if (mt.uniform(0, 1) < a->background_frequency) {
        bg_vol = mt.uniform(0, a->background_volume_range);
      } else if (sil && mt.uniform(0, 1) < 0.9) {
        bg_vol = mt.uniform(0, a->silence_volume_range);
      }
    }
    double fg;
    if (sil) {
      fg = 0.0;
    } else {
      fg = 1.0;
      if (mt.uniform(0, 1) < a->foreground_frequency) fg = 1.0 + mt.uniform(-a->foreground_volume_range, a->foreground_volume_range);
      if (mt.uniform(0, 1) < a->flip_frequency) fg *= -1.0;
    }
    out_rows[k] = row; out_labels[k] = lab; out_shift[k] = shift; out_bg_off[k] = bg_off;
    out_bg_vol[k] = (float)bg_vol; out_fg_vol[k] = (float)fg;
  }
  *mt_pos = mt.pos;
  return KWS_OK;
}
