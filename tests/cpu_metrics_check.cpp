// Host build of layout_dm_amd/csrc/ldm_layout_metrics_core.h (the per-element arithmetic of kernels_metrics.hip):
// reads {int32 B, int32 S, float bbox[B][S][4], uint8 mask[B][S]} and writes float out[B][6].  tests/test_layout_metrics.py
// runs it against the reference-produced fixture and the oracle restatement.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../layout_dm_amd/csrc/ldm_layout_metrics_core.h"

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[2];
  if (fread(hdr, 4, 2, f) != 2) return 2;
  const int B = hdr[0], S = hdr[1];
  std::vector<float> bbox((size_t)B * S * 4), out((size_t)B * ldm_metrics::kNumMetrics);
  std::vector<uint8_t> mask((size_t)B * S);
  if (fread(bbox.data(), 4, bbox.size(), f) != bbox.size() || fread(mask.data(), 1, mask.size(), f) != mask.size()) return 2;
  fclose(f);
  for (int b = 0; b < B; ++b) {
    const float* bb = bbox.data() + (size_t)b * S * 4;
    const uint8_t* mm = mask.data() + (size_t)b * S;
    int nv = 0;
    for (int i = 0; i < S; ++i) nv += mm[i] ? 1 : 0;
    ldm_metrics::layout_scores(S, nv, [&](int i) { return ldm_metrics::element_terms(bb, mm, S, i); },
                               out.data() + (size_t)b * ldm_metrics::kNumMetrics);
  }
  f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(out.data(), 4, out.size(), f);
  fclose(f);
  return 0;
}
