// Environment knobs of libldm_hip.so, in ONE place.
//
// Every knob selects a development path: a fallback / older kernel for a same-box A/B, a tuning override, or a timing
// ablation (some of which produce wrong numerics by design).  None is needed in production, and a stray variable must
// not be able to change the benchmarked path silently: knobs are honoured only when LDM_DEV=1 is set as well —
// otherwise ldm_create REFUSES to build a handle while one of them is present in the environment, and the launchers
// ignore them.  What was honoured is reported by ldm_describe (bench.py records it from there, not from os.environ).
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace ldm {

// name -> what it selects
inline const std::map<std::string, const char*>& knob_table() {
  static const std::map<std::string, const char*> t = {
      {"LDM_CHUNK", "layouts per pass of the per-step path"},
      {"LDM_LANES", "concurrent chunk pipelines of the per-step path"},
      {"LDM_LANE_OFFSET_US", "phase offset between lanes"},
      {"LDM_GEMM_CFG", "tile configuration ids of the generic fp16 GEMMs"},
      {"LDM_FUSED_ATTN", "0 = generic tiled kernels instead of the layout-resident stack kernel"},
      {"LDM_STACK_LOOP", "0 = per-step launches instead of the one-launch reverse loop"},
      {"LDM_POST_WAVE", "wavefront-per-token step tail instead of the 16-lane groups"},
      {"LDM_ATTN32", "rows|staged: older fp32 attention kernels; direct: the fp32-MFMA kernel in the split mode too"},
      {"LDM_ATTN_ABL", "fp16 attention timing ablations (WRONG NUMERICS)"},
      {"LDM_ATTN_TM", "stack kernel phase-timer instantiation"},
      {"LDM_GEMM32_WIDE", "160-wide fp32 GEMM tiles for N = 464"},
      {"LDM_GEMM32_SLOTS", "resident fp32 GEMM workgroups per CU"},
      {"LDM_GEMM32_BM64", "64-row fp32 GEMM tiles for one-round shapes"},
      {"LDM_REL_FUSED", "0 = three launches per cond=relation step of the per-step path (pre-r04 structure)"},
      {"LDM_X3_CFG", "tile configuration of the split GEMM (8: 256x256 default, 0: 128x128, 1: 4 stages, 2: 256x128, 3: 128x256, 5: 128x128x64, 6 / 7: operands through registers, 9: dependent MFMA order, 10 / 11: non-temporal activation fills in the two-reader GEMMs / in every GEMM)"},
      {"LDM_X3_GRP", "column-group width of the split GEMM's tile order (0 = row-major)"},
      {"LDM_X3_LNGEMM", "0 = LayerNorm launches + gemm16x3_k for the LayerNorm-fed GEMMs of the split mode instead of the row-resident LayerNorm + x3 GEMM (the r04 structure); 1 = the row-resident kernels without a GEMM prologue; 2 / 3 = out_proj too / only out_proj as a GEMM prologue (default 4: linear2 only)"},
      {"LDM_X3_ATTNOUT", "0 = attn16x3_k + the out_proj launch of gemm16x3_k instead of the fused layout-resident attention + out_proj kernel of the split mode (the r05 structure)"},
      {"LDM_X3_HIDPANEL", "0 = row-major hi / lo hidden activations between linear1 and the linear2 GEMM prologue instead of the panel-major form (the r05 layout)"},
      {"LDM_HYB_FFN", "0 = the hybrid mode's FFN as two launches (linear1 writing plain-fp16 panels, linear2 as the next launch's GEMM prologue) instead of the fused plain-fp16 FFN kernel"},
      {"LDM_BALANCED_CHUNKS", "0 = a call's passes are full chunks plus a remainder (300 layouts = 256 + 44) instead of even shares (150 + 150)"},
      {"LDM_HYB_ATTNFFN", "0 = the hybrid mode's fused fp16 FFN as its own launch (kernels_ffn16.hip) instead of behind the attention in the attention launch"},
      {"LDM_ATTNOUT_TM", "fused attention + out_proj kernel: phase-timer instantiation (tools/attnout_probe.py phases)"},
      {"LDM_LNGEMM_TM", "row-resident LayerNorm + x3 GEMM: phase-timer instantiation (tools/lngemm_probe.py)"},
      {"LDM_LNGEMM_ABL", "row-resident LayerNorm + x3 GEMM: compile-time timing variants, measurement build only (tools/build_measurement_variants.py lngemm; WRONG NUMERICS)"},
      {"LDM_SPLIT_GEMM", "old = the register-staged split GEMM of r03 instead of the LDS-DMA fp16 x 3 kernel"},
      {"LDM_REL_LOOP", "0 = cond=relation on the per-step path instead of inside the one-launch loop (fast mode)"},
  };
  return t;
}

struct KnobState {
  std::mutex mu;
  std::map<std::string, std::string> honoured;
};
inline KnobState& knob_state() {
  static KnobState s;
  return s;
}

inline bool knob_dev_mode() {
  const char* v = getenv("LDM_DEV");
  return v && atoi(v) != 0;
}

// value of a knob, or nullptr when it is unset or the library is not in dev mode
inline const char* knob_env(const char* name) {
  const char* v = getenv(name);
  if (!v || !knob_dev_mode()) return nullptr;
  KnobState& s = knob_state();
  std::lock_guard<std::mutex> lk(s.mu);
  s.honoured[name] = v;
  return v;
}
inline int knob_int(const char* name, int dflt) {
  const char* v = knob_env(name);
  return v ? atoi(v) : dflt;
}

// first knob present in the environment although LDM_DEV is not set ("" = none): ldm_create refuses on it
inline std::string knob_refused() {
  if (knob_dev_mode()) return "";
  for (const auto& kv : knob_table())
    if (getenv(kv.first.c_str())) return kv.first;
  return "";
}

// "LDM_DEV=1;LDM_STACK_LOOP=0;..." — the knobs that have been honoured so far in this process
inline std::string knobs_honoured() {
  KnobState& s = knob_state();
  std::lock_guard<std::mutex> lk(s.mu);
  std::string out = knob_dev_mode() ? "LDM_DEV=1" : "";
  for (const auto& kv : s.honoured) out += (out.empty() ? "" : ";") + kv.first + "=" + kv.second;
  return out;
}

}  // namespace ldm
