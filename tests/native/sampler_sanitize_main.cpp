// Driver for the AddressSanitizer / UBSan build of the host-side sampler (tests/test_sanitizers_cpu.py): exercises
// kws_sampler_draw over random and edge-case arguments with tightly sized heap buffers, so any out-of-bounds access,
// signed overflow or misaligned access inside csrc/sampler.cpp aborts the process.  Test infrastructure only.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/kws_hip.h"

static char g_err[512];
void kws_set_error(const char* fmt, ...) { snprintf(g_err, sizeof(g_err), "%s", fmt); }   // the library's sink, stubbed

static uint32_t lcg(uint32_t& s) { return s = s * 1664525u + 1013904223u; }

int main() {
  uint32_t seed = 12345;
  int ok = 0, refused = 0;
  for (int it = 0; it < 400; ++it) {
    const int n_cand = 1 + lcg(seed) % 50, n_pseudo = lcg(seed) % 20;
    std::vector<int32_t> rows(n_cand), labs(n_cand), prow(n_pseudo ? n_pseudo : 1), plab(n_pseudo ? n_pseudo : 1);
    std::vector<uint8_t> sil(n_cand), psil(n_pseudo ? n_pseudo : 1);
    for (int i = 0; i < n_cand; ++i) { rows[i] = lcg(seed) % 1000; labs[i] = lcg(seed) % 12; sil[i] = lcg(seed) % 7 == 0; }
    for (int i = 0; i < n_pseudo; ++i) { prow[i] = lcg(seed) % 1000; plab[i] = lcg(seed) % 12; psil[i] = 0; }
    kws_sampler_set_t cand{rows.data(), labs.data(), sil.data(), n_cand};
    kws_sampler_set_t pseudo{prow.data(), plab.data(), psil.data(), n_pseudo};
    const int n_bg = lcg(seed) % 4;
    std::vector<int64_t> bg_len(n_bg ? n_bg : 1), bg_start(n_bg ? n_bg : 1);
    int64_t acc = 0;
    for (int i = 0; i < n_bg; ++i) {
      bg_len[i] = (it % 9 == 0 && i == 0) ? 16000 : 16001 + lcg(seed) % 100000;   // some too short: must be refused
      bg_start[i] = acc;
      acc += bg_len[i];
    }
    kws_sampler_args_t a;
    memset(&a, 0, sizeof(a));
    a.deterministic = lcg(seed) % 3 == 0;
    a.count = lcg(seed) % 40;
    a.offset = a.deterministic ? (int)(lcg(seed) % (n_cand + 1)) : 0;
    if (a.deterministic && a.offset + a.count > n_cand) a.count = n_cand - a.offset;
    a.use_background = n_bg > 0 && !a.deterministic;
    a.n_bg = n_bg; a.bg_len = bg_len.data(); a.bg_start = bg_start.data();
    a.desired_samples = 16000;
    a.shift_lo = -(int)(lcg(seed) % 600); a.shift_hi = (it % 11 == 0) ? a.shift_lo - 1 : (int)(lcg(seed) % 3);
    a.background_frequency = 0.3; a.background_volume_range = 0.15; a.foreground_frequency = 0.3;
    a.foreground_volume_range = 0.15; a.time_shift_frequency = 0.5; a.pseudo_frequency = n_pseudo ? 0.6 : 0.0;
    a.flip_frequency = 0.1; a.silence_volume_range = 0.3;
    std::vector<uint32_t> key(624);
    for (auto& k : key) k = lcg(seed);
    int pos = lcg(seed) % 625;
    const int n = a.count > 0 ? a.count : 1;
    std::vector<int32_t> o_rows(n), o_lab(n), o_shift(n);
    std::vector<int64_t> o_off(n);
    std::vector<float> o_bgv(n), o_fgv(n);
    const int rc = kws_sampler_draw(key.data(), &pos, &cand, &pseudo, &a, o_rows.data(), o_lab.data(), o_shift.data(),
                                    o_off.data(), o_bgv.data(), o_fgv.data());
    if (rc == KWS_OK) {
      ++ok;
      for (int i = 0; i < a.count; ++i) {
        if (o_lab[i] < 0 || o_lab[i] >= 12 || o_shift[i] < a.shift_lo || o_shift[i] > (a.shift_hi > 0 ? a.shift_hi : 0)) return 2;
        if (a.use_background && (o_off[i] < 0 || o_off[i] + 16000 > acc)) return 3;    // the noise slice stays inside the buffer
      }
    } else {
      ++refused;
    }
    if (pos < 0 || pos > 624) return 4;
  }
  // NULL arguments are refused, not dereferenced
  if (kws_sampler_draw(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == KWS_OK)
    return 5;
  printf("sampler under ASan+UBSan: %d draws ok, %d refused\n", ok, refused);
  return (ok > 100 && refused > 10) ? 0 : 6;
}
