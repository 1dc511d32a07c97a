// Internal interface between the C-ABI entry point of the fused chain (mvsn_chain.hip) and its two kernels:
// the direct-form kernel (any coarse grid up to 2048 px) and the Winograd F(2x2,3x3) kernel (even grids whose
// activation planes + one layer of transformed weights fit the 160 KB of LDS: 16x32 at 512x256 frames).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mvsn {

constexpr int CH_W0_FLOATS = 9 * 9 * 2 * 64;  // direct form, conv0: 35 -> 36 input channels = 9 k-steps per tap
constexpr int CH_W1_FLOATS = 9 * 8 * 2 * 64;
constexpr int CH_SP_FLOATS = 7 * 32;          // biases and GroupNorm affine
constexpr int CH_DIRECT_FLOATS = CH_W0_FLOATS + 2 * CH_W1_FLOATS + CH_SP_FLOATS;
// Winograd form: U = G g G^T per (cout, cin), [k-step of 4 cin][cout tile][xi quad][lane][4 xi] = 2048 floats per k-step
constexpr int CW_UCHUNK = 2 * 4 * 64 * 4;
constexpr int CW_U0_FLOATS = 9 * CW_UCHUNK;
constexpr int CW_U1_FLOATS = 8 * CW_UCHUNK;
// stepwise form: the three layers in the convolution kernels' own Winograd layout (mvsn_conv_wino.hip),
// [chunk of 4 cin][xi][cout tile][lane] = 2048 floats per chunk
constexpr int CH_STEPS_OFFSET = CH_DIRECT_FLOATS + CW_U0_FLOATS + 2 * CW_U1_FLOATS;
constexpr int CS_U0_FLOATS = 9 * 2048, CS_U1_FLOATS = 8 * 2048;
constexpr int CH_PACKED_FLOATS = CH_STEPS_OFFSET + CS_U0_FLOATS + 2 * CS_U1_FLOATS;

struct ChainArgs {
  const float *src;      // (N,3,P)
  const float *H;        // (N,D,9)
  const float *Hinc;     // (N,D,9)
  const float *f0;       // (N,32,P)
  const float *fl;       // (B,32,P)
  const float *packed;   // CH_PACKED_FLOATS
  float *cost;           // (N,32,D,P)
  uint8_t *mask;         // (N,D,P)
  float *fvol;           // (N,32,D,P) or null
  float *workspace;      // global activation planes or null
  int B, D, rows, cols, CS;
  int chain0;            // banded form, launched in passes: global index of this pass's first chain (0 otherwise)
  int ws_chains;         // ... and the number of chains its workspace holds (the status word sits behind them)
  unsigned long long *dbg;  // optional: s_memtime stamps of block 0 / lane 0 at phase boundaries (tuning only)
  // Repair launch (mvsn_incremental_cost_volume_guarded): `gate` = the status word the banded launch ahead of this one
  // left behind; the launch runs only if it is non-zero (a hand-off timed out) and then recomputes every output.
  // `sticky` (optional, device-visible -- e.g. pinned host memory): [0] |= status, [1] += 1 per repair that ran
  // (one writer per launch; not atomic across concurrent streams -- a lost count still leaves a changed word).
  const unsigned *gate;
  unsigned *sticky;
  // bf16 feature tier (BASELINE config 5; never the parity path): `cost` points at (N,32,D,P) bf16 elements and the
  // kernels store the same cost values rounded to nearest-even (template C16 of the kernels that have the variant)
  int cost_bf16;
};

// ---- cost-volume stores, by element type of the volume (float: the contract; uint16_t: bf16 bits, C16 kernels) ----
typedef float chain_f2 __attribute__((ext_vector_type(2)));
typedef float chain_f4 __attribute__((ext_vector_type(4)));
typedef unsigned chain_u2 __attribute__((ext_vector_type(2)));
template <bool C16> struct ChainCost { typedef float type; };
template <> struct ChainCost<true> { typedef uint16_t type; };
__device__ __forceinline__ unsigned chain_bf16_pair(float a, float b) {   // two RNE conversions: one v_cvt_pk_bf16_f32
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 p;
  p[0] = (__bf16)a;
  p[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void chain_cost_nt(float *p, chain_f2 v) { __builtin_nontemporal_store(v, reinterpret_cast<chain_f2 *>(p)); }
__device__ __forceinline__ void chain_cost_nt(uint16_t *p, chain_f2 v) {
  __builtin_nontemporal_store(chain_bf16_pair(v.x, v.y), reinterpret_cast<unsigned *>(p));
}
__device__ __forceinline__ void chain_cost_nt(float *p, chain_f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<chain_f4 *>(p)); }
__device__ __forceinline__ void chain_cost_nt(uint16_t *p, chain_f4 v) {
  chain_u2 q;
  q.x = chain_bf16_pair(v.x, v.y), q.y = chain_bf16_pair(v.z, v.w);
  __builtin_nontemporal_store(q, reinterpret_cast<chain_u2 *>(p));
}
__device__ __forceinline__ void chain_cost_st(float *p, chain_f4 v) { *reinterpret_cast<chain_f4 *>(p) = v; }
__device__ __forceinline__ void chain_cost_st(uint16_t *p, chain_f4 v) {
  chain_u2 q;
  q.x = chain_bf16_pair(v.x, v.y), q.y = chain_bf16_pair(v.z, v.w);
  *reinterpret_cast<chain_u2 *>(p) = q;
}
__device__ __forceinline__ void chain_cost_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void chain_cost_st(uint16_t *p, float v) { *p = (uint16_t)(chain_bf16_pair(v, 0.0f) & 0xffffu); }

// First statement of a kernel that may be launched as a repair: true = nothing to repair, the workgroup returns.
// (The status word is read past L1 -- the launch boundary published it, a stale line of an earlier forward must not
// answer; every thread reads the same word, so the answer is workgroup-uniform.)
__device__ __forceinline__ bool chain_gate_closed(const ChainArgs &a) {
  if (a.gate == nullptr) return false;
  const unsigned st = __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (st == 0) return true;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.sticky != nullptr) {
    // one writer per launch, launches of a stream in order: plain system-scope loads / stores (a read-modify-write
    // atomic on pinned HOST memory would need PCIe atomics end to end)
    const unsigned seen = __hip_atomic_load(a.sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned count = __hip_atomic_load(a.sticky + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.sticky, seen | st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.sticky + 1, count + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return false;
}

// the chain's buffers as plain pointer arguments behind the struct (MVSN_VIS10, mvsn_common.h)
#define CHAIN_VISIBLE(a)                                                                                              \
  (const void *)(a).src, (const void *)(a).H, (const void *)(a).Hinc, (const void *)(a).f0, (const void *)(a).fl,     \
      (const void *)(a).packed, (const void *)(a).cost, (const void *)(a).mask, (const void *)(a).fvol,               \
      (const void *)(a).workspace

// ... of a launch that may be a repair: the gate word (inside the banded launch's workspace) takes the slot of this
// launch's own workspace, which is scratch within the launch and orders nothing between kernels
#define CHAIN_VISIBLE_G(a)                                                                                            \
  (const void *)(a).src, (const void *)(a).H, (const void *)(a).Hinc, (const void *)(a).f0, (const void *)(a).fl,     \
      (const void *)(a).packed, (const void *)(a).cost, (const void *)(a).mask, (const void *)(a).fvol,               \
      ((a).gate ? (const void *)(a).gate : (const void *)(a).workspace)

// Winograd form: does this coarse grid have a plan (even rows / cols, <= 128 patches, LDS fits)?
bool chain_wino_supported(int rows, int cols);
int chain_wino_launch(const ChainArgs &a, int n_chains, hipStream_t stream);

// Stepwise form (mvsn_chain_steps.hip): one plane per round of full-chip launches
bool chain_steps_supported(int rows, int cols);
size_t chain_steps_workspace_bytes(int n_chains, int D, int rows, int cols);
int chain_steps_launch(const ChainArgs &a, int n_chains, void *workspace, size_t workspace_bytes, hipStream_t stream);

// Banded form (mvsn_chain_band.hip): one chain on several workgroups; coarse grids 16x32, 30x40, 32x64
bool chain_band_supported(int rows, int cols);
int chain_band_groups(int n_chains, int rows, int cols);    // workgroups per chain (0: no plan for this grid)
int chain_band_chains_per_pass(int rows, int cols);        // chains whose workgroups are co-resident (one per CU)
size_t chain_band_workspace_bytes(int n_chains, int rows, int cols);
size_t chain_band_status_offset(int n_chains, int rows, int cols);
void chain_band_debug_flags(int flags);   // test hook, see mvsn_debug_set_band_flags
int chain_band_launch(const ChainArgs &a, int n_chains, void *workspace, size_t workspace_bytes, int flags,
                      hipStream_t stream);

// Slab plans of the banded form (mvsn_chain_slab.hip): a few fat bands per chain, 512-thread workgroups -- what the
// banded dispatcher launches once more chains are in flight than ONE pass of the thin-band plan holds
struct SlabPlan {
  int G, threads;
  size_t chain_u64, lds_bytes;
  void (*kernel)(ChainArgs, int, const void *, const void *, const void *, const void *, const void *, const void *,
                 const void *, const void *, const void *, const void *);
  void (*kernel16)(ChainArgs, int, const void *, const void *, const void *, const void *, const void *, const void *,
                   const void *, const void *, const void *, const void *);   // ... storing the cost volume as bf16
};
bool chain_slab_plan(int rows, int cols, SlabPlan *p);

}  // namespace mvsn
