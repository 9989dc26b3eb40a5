// Host build of layout_dm_amd/csrc/ldm_post_token.h (the per-token scalar tail of a reverse step, the form a lane of the
// stack kernel runs behind the fused head): reads a case file written by tests/test_post_token_scalar.py, writes the
// drawn tokens.  File layout (little endian): int32 header[10] = {magic, N, C, pad_id, mask_id, kind, top_k, f64_lse,
// has_weak, alias_work}; float temperature, top_p; uint64 seed; then per-token arrays int32 tok, start, count, cond_tok, strong,
// pad_disable, pos, step [N each]; uint64 layout[N]; float sched[N][10]; float logits[N][C]; float weak[N][C] if has_weak.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../layout_dm_amd/csrc/ldm_post_token.h"

template <typename T>
static std::vector<T> rd(FILE* f, size_t n) {
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  const auto hdr = rd<int32_t>(f, 10);
  if (hdr[0] != 0x4C444D31) return 3;
  const int N = hdr[1], C = hdr[2];
  const auto fl = rd<float>(f, 2);
  const auto seed = rd<uint64_t>(f, 1);
  const auto tok = rd<int32_t>(f, N), start = rd<int32_t>(f, N), count = rd<int32_t>(f, N), cond_tok = rd<int32_t>(f, N),
             strong = rd<int32_t>(f, N), pad_dis = rd<int32_t>(f, N), pos = rd<int32_t>(f, N), step = rd<int32_t>(f, N);
  const auto layout = rd<uint64_t>(f, N);
  const auto sched = rd<float>(f, (size_t)N * 10);
  const auto logits = rd<float>(f, (size_t)N * C);
  const auto weak = rd<float>(f, hdr[8] ? (size_t)N * C : 0);
  fclose(f);
  std::vector<int32_t> out(N);
  for (int i = 0; i < N; ++i) {
    ldm_post::TokenArgs a{};
    a.logits = &logits[(size_t)i * C];
    a.tok = tok[i]; a.start = start[i]; a.count = count[i];
    a.pad_id = hdr[3]; a.mask_id = hdr[4]; a.n_class = C;
    a.cond_tok = cond_tok[i]; a.strong = strong[i] != 0;
    a.weak = hdr[8] ? &weak[(size_t)i * C] : nullptr;
    a.weak_stride = 1;
    a.pad_disable = pad_dis[i] != 0;
    a.kind = hdr[5]; a.temperature = fl[0]; a.top_p = fl[1]; a.top_k = hdr[6];
    a.pos = (uint32_t)pos[i]; a.step = (uint32_t)step[i]; a.layout = layout[i]; a.seed = seed[0];
    const float* s = &sched[(size_t)i * 10];
    const ldm_post::StepSchedule sc{s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9]};
    // working storage: separate, or (alias_work) the token's own logits row, as a lane of the kernel uses it
    std::vector<float> row(logits.begin() + (size_t)i * C, logits.begin() + (size_t)(i + 1) * C);
    std::vector<float> sep(3 * (size_t)(count[i] + 2));
    if (hdr[9]) {
      if ((int)row.size() < 3 * (count[i] + 2)) return 4;
      a.logits = row.data();
    }
    float* work = hdr[9] ? row.data() : sep.data();
    out[i] = hdr[7] ? ldm_post::step_token<true>(a, sc, work) : ldm_post::step_token<false>(a, sc, work);
  }
  FILE* o = fopen(argv[2], "wb");
  if (!o) return 1;
  fwrite(out.data(), sizeof(int32_t), N, o);
  fclose(o);
  return 0;
}
