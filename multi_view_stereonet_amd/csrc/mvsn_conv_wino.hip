// Winograd F(2x2, 3x3) form of the 2-D 3x3 stride-1 convolutions with 32 output channels and of the 3x3x3
// 32 -> 32 layers (template VOL: 2-D Winograd products summed over the depth tap)
// (mvsn_conv_forward / mvsn_conv_forward_blocks with desc.precision = MVSN_CONV_FP32_WINO).
//
// The fp32 matrix pipe is the ceiling of the refiner layers (v_mfma_f32_16x16x4_f32 at 1/16 of the bf16 rate,
// and the direct kernels already keep it ~85 % busy at the sustained clock), so the remaining lever is fewer
// multiplies: Y = A^T [ (G g G^T) . (B^T d B) ] A turns every 2x2 output patch into 16 products per
// (cin, cout) pair instead of 36.  The sum over input channels is taken in the transformed domain, i.e. one
// small GEMM per Winograd coefficient xi:  M_xi[patch][cout] = sum_cin V_xi[patch][cin] * U_xi[cin][cout].
// All arithmetic is fp32; the result differs from the direct form by rounding only (~1e-6 relative).
//
// Persistent workgroups, one per CU: 512 threads (8 waves), a 16 x 32 output tile (8 x 16 patches) at a time:
//   * the layer's transformed weights U (16 coefficients x cin x 32 couts, 64 KB at 32 input channels) are
//     loaded into LDS ONCE per workgroup and stay there while it walks its tiles -- streamed per chunk they
//     were two thirds of the DMA instructions, and the CU's DMA issue rate (~1 piece per ~115 cycles), not the
//     matrix pipe, set the pace;
//   * the haloed raw tiles (dilation 1: 18 rows x 40 columns from the aligned column x0 - 4) arrive by LDS-DMA
//     (16-byte pieces, 12 instructions per 4-channel chunk per CU) in a 3-6 stage ring that runs ahead across tile
//     boundaries (a chunk is only 32 MFMAs per wave, shorter than a DMA round trip); one barrier per step of one
//     or two chunks (template KS).  Round 4: the pieces are `buffer_load_dwordx4 .. offen lds` on a descriptor of the
//     channel plane -- one 32-bit byte offset per piece, computed once per tile; pieces outside the image carry
//     0xFFFFFFFF and the hardware's range check writes their zeros (wn_dma16_buf);
//   * wave w owns patch row w (16 patches) and both cout tiles.  Lane (k = lane>>4, p = lane&15) reads the 4 x 4
//     input patch p of channel k from the raw tile and transforms it in registers (B^T d B: 32 adds): the 16
//     coefficients it ends up with are exactly its A-fragment values (A = V_xi: 16 patches x 4 cins), so the
//     transformed input never touches LDS;
//   * per chunk 16 xi x 2 MFMAs with B = U_xi (4 cins x 16 couts) read from LDS; 128 accumulator registers;
//   * output transform in registers: A^T m A (24 adds per patch), bias, per-wave GroupNorm partials.  Dilated
//     layers: D = patches x couts (a lane holds 4 consecutive patches of one cout = 8 output columns of 2 rows,
//     16-byte stores); dilation 1 (round 4): operands swapped, D = couts x patches (a lane holds 4 couts of ONE patch,
//     float2 stores: 16 lanes = 128 contiguous bytes of an output row).  All stores through buffer descriptors.
//   * (round 4) per k-step half: the 32 MFMAs as one burst, then the next step's input transform as one block
//     (MVSN_WN_XF); every VALU instruction next to the multiplies costs the matrix pipe ~2 cycles per MFMA
//     (profiles/r04_micro/README.md), so address arithmetic lives in descriptors and scalar registers.
// MODE 1: the previous layer's LeakyReLU(GroupNorm(.)) is applied in LDS, once per element, by the wave that
// fetched the piece (out-of-image pieces keep their zeros, as the padding of the materialised tensor would be).
// LDS layout details that matter: dilation-1 tiles sit one float further (patches start at even columns: a row is
// one ds_read2_b64 over all banks), odd channels of dilated tiles DIL floats further (conflict-free strided reads);
// GroupNorm partials are one record per (wave, 16-lane row), reduced with DPP adds.  DESIGN.md sections 3.2b / 3.2c
// have the measurements behind each of these choices, section 3.6 the variants that lost.
#include <type_traits>

#include "mvsn_common.h"
#include "mvsn_conv_wino.h"

namespace mvsn {

constexpr int WN_THREADS = 512, WN_WAVES = 8;
constexpr int WN_TY = 16, WN_TX = 32;                  // output tile
// haloed raw tile of a layer with dilation DIL (a dilated layer is DIL x DIL interleaved dilation-1 problems:
// a "2 x 2 patch" is the outputs (y, x), (y, x + DIL), (y + DIL, x), (y + DIL, x + DIL), its input window the
// 4 x 4 samples at stride DIL)
constexpr int wn_pa(int dil) { return (dil + 3) / 4 * 4; }                 // halo rounded up to 16-byte columns
constexpr int wn_xs(int dil) { return WN_TX + 2 * wn_pa(dil); }            // row stride: columns x0 - pa .. x0 + 31 + pa
constexpr int wn_hy(int dil) { return WN_TY + 2 * dil; }                   // haloed rows
constexpr int wn_groups(int dil) { return wn_hy(dil) * (wn_xs(dil) / 4); } // 16-byte groups per channel tile
constexpr int wn_pieces(int dil) { return (wn_groups(dil) + 63) / 64; }    // DMA instructions per channel
constexpr int wn_rcst(int dil) { return wn_hy(dil) * wn_xs(dil) + 16; }    // raw channel stride (floats)
constexpr int WN_UFLOATS = 16 * 128;                   // U fragments per chunk: [xi][cout tile][lane]
constexpr int WN_MAX_CHUNKS = 9;                       // resident U: up to 36 input channels (72 KB)

__device__ floatx4 g_wn_zero16 = {0.f, 0.f, 0.f, 0.f};
#define WN_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define WN_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

#ifndef MVSN_WN_ABLATE   // tuning aid: bit 0 no input transform, 1 no U fragment reads, 2 no DMA in the loop, 3 no barrier, 4 no epilogue, 5 no raw reads, 6 no output stores
#define MVSN_WN_ABLATE 0
#endif

#ifndef MVSN_RD_ABLATE   // tuning aid, carried jobs: bit 0 no loads, 1 no arithmetic / stores, 2 no residual fetch
#define MVSN_RD_ABLATE 0
#endif

#ifndef MVSN_WN_SHIFT
#define MVSN_WN_SHIFT 1
#endif
#ifndef MVSN_WN_NT_STORES   // tuning aid: output tiles by non-temporal stores
#define MVSN_WN_NT_STORES 0
#endif
#ifndef MVSN_WN_TRANSPOSED   // dilation-1 kernels: D = couts x patches (0: the patches x couts form of the dilated layers)
#define MVSN_WN_TRANSPOSED 1
#endif
#ifndef MVSN_WN_BUFDMA    // raw tiles through a buffer descriptor (hardware range check), 0: flat addresses + zero line
#define MVSN_WN_BUFDMA 1
#endif
#ifndef MVSN_WN_PIN       // input-transform results pinned where the source computes them (see multiply())
#define MVSN_WN_PIN 1
#endif
#ifndef MVSN_WN_NT_TR     // tuning aid: the transposed epilogue's (full-line) stores non-temporal
#define MVSN_WN_NT_TR 0
#endif
#ifndef MVSN_WN_NT_RAW    // tuning aid: raw-tile DMA with the nt hint
#define MVSN_WN_NT_RAW 0
#endif
#ifndef MVSN_WN_XF        // placement of the next step's input transform, see conv_wino_kernel's multiply()
#define MVSN_WN_XF 2
#endif
#ifndef MVSN_WN_STAGGER   // tuning aid: waves 4-7 sleep this many 64-cycle units behind every step barrier
#define MVSN_WN_STAGGER 0
#endif

#ifdef MVSN_WN_STAMPS   // tuning aid (tools/wino_phases.py): s_memtime stamps of one mid-launch wave
__device__ unsigned long long *g_wn_stamps = nullptr;
// (kept in LDS and copied out at the end: a global store per stamp would sit in the vmcnt queue the kernel waits on)
#define WN_STAMP() do { if (dbg && dbg_on && dbg_i < 120) dbg_lds[dbg_i++] = __builtin_readcyclecounter(); } while (0)
#else
#define WN_STAMP() do { } while (0)
#endif

// Workgroup barrier that publishes LDS traffic only (LDSONLY): __syncthreads() is a release fence and the compiler puts
// s_waitcnt vmcnt(0) in front of it -- every step waits for ALL DMA pieces and output stores in flight.  A wave's own
// pieces are covered by wait_landed.  (With the builtin DMA below the compiler moves that drain behind the barrier
// instead; measured per instantiation, see conv_wino_kernel.)
template <bool LDSONLY>
__device__ __forceinline__ void wn_barrier() {
  if constexpr (LDSONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

// One LDS-DMA piece: lane i's 16 bytes at g go to l + 16 i (l wave-uniform).  ASM: issued as inline assembly -- the
// compiler orders every LDS read behind a __builtin_amdgcn_global_load_lds it has seen with s_waitcnt vmcnt(0) (it
// cannot tell the ring stage being filled from the one being read); in the kernels that carry a pass this drain sits
// right behind the pass's freshly issued loads and exposes their whole latency every step.  The waits that matter are
// wait_landed's counted ones.  (M0 = LDS address; one wait state between the M0 write and the DMA.)  The builtin form
// schedules better where the drain is harmless (measured on MI355X, level-0 layer of 64 images: plain dilation-1
// layer 0.79 ms builtin / 0.81 asm; the same layer carrying a pass 1.14 / 1.06, with the input transform 1.29 / 1.18;
// dilation 2, 4, 8 within 1 % either way).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (M0 is a reserved register; the compiler keeps nothing in it here)
template <bool ASM>
__device__ __forceinline__ void wn_dma16(const float *g, const float *l) {
  if constexpr (ASM) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)WN_LPTR(l));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory", "m0");
  } else {
    __builtin_amdgcn_global_load_lds(WN_GPTR(g), WN_LPTR(l), 16, 0, 0);
  }
}
// The same piece through a buffer descriptor (base = a channel plane, num_records = its bytes): lanes whose offset lies
// outside -- the marker 0xFFFFFFFF of pieces outside the image, or every lane when the descriptor is empty (a channel /
// plane that does not exist) -- get ZEROS written to their LDS slot by the hardware's range check.  The flat form needed
// a 64-bit address per lane and piece with two selects against a zero line: ~8 VALU instructions per piece and step.
template <bool ASM>
__device__ __forceinline__ void wn_dma16_buf(const float *base, unsigned bytes, unsigned voff, const float *l) {
  if constexpr (ASM) {
    typedef int wn_srd_t __attribute__((ext_vector_type(4)));
    const size_t b = (size_t)base;
    wn_srd_t srd;
    srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    srd[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    srd[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    srd[3] = 0x00020000;
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)WN_LPTR(l));
#if MVSN_WN_NT_RAW
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds" ::"v"(voff), "s"(srd), "s"(la)
                 : "memory", "m0");
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(srd), "s"(la)
                 : "memory", "m0");
#endif
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000),
                                             WN_LPTR(l), 16, (int)voff, 0, 0, MVSN_WN_NT_RAW ? 2 : 0);
  }
}
#pragma clang diagnostic pop

// Output tiles and the carried job go through buffer descriptors too: the base and the per-(cout, row) part of an
// address are scalar (descriptor + soffset), the lane's part is ONE 32-bit offset formed once per tile, and a lane whose
// columns lie outside the image carries the offset 0xFFFFFFFF -- the range check drops its store.  The flat form spent
// ~10 VALU instructions per store on 64-bit address arithmetic, 16 stores per tile and wave, each behind its own
// exec-mask branch: a quarter of a 2-D tile's VALU instructions.
typedef unsigned wn_uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned wn_uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wn_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
template <int AUX>   // 2 = non-temporal
__device__ __forceinline__ void wn_store2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float a, float b) {
  wn_uintx2 d;
  d[0] = __builtin_bit_cast(unsigned, a), d[1] = __builtin_bit_cast(unsigned, b);
  __builtin_amdgcn_raw_buffer_store_b64(d, r, (int)voff, (int)soff, AUX);
}
template <int AUX>   // 2 = non-temporal
__device__ __forceinline__ void wn_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, floatx4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wn_uintx4, v), r, (int)voff, (int)soff, AUX);
}

struct WinoDiv {
  unsigned mul, shift;
};
static WinoDiv wino_div(unsigned d) {
  unsigned s = 0;
  while ((1u << s) < d) ++s;
  const unsigned long long m = ((1ull << 32) * ((1ull << s) - d)) / d + 1;
  return WinoDiv{(unsigned)m, s};
}
__device__ __forceinline__ int wdiv(int n, WinoDiv f) { return (int)((__umulhi((unsigned)n, f.mul) + (unsigned)n) >> f.shift); }

struct WinoArgs {
  int n, cin, H, W, ntx, tiles, nchunks;
  WinoDiv fd_ntx;
  // input as up to three channel blocks (each a contiguous (n, c_b, H, W) tensor): channels [0, cb0) from `in`,
  // [cb0, cb0 + cb1) from in1, the rest from in2.  One block: cb0 = cin.
  int cb0, cb1;
  const float *in1, *in2;
  size_t bs0, cs0;   // block 0: floats between samples / between channels (cb0 * plane, plane when contiguous)
  int rev;           // walk the work items from the last to the first (placement only: results are identical)
  int D;   // planes per sample (volume form; 1 for the 2-D layers)
  int pr;  // WIDE == 2: patch rows per plane, (H + 1) / 2 (there `tiles` = work items per SAMPLE, see conv_wino_kernel)
  WinoDiv fd_pr;
};

// A normalise / activate / add pass over ANOTHER tensor that this launch's waves carry along (template RIDE = 256-float
// units per wave and step): out = LReLU(GN(x)) [+ residual | + LReLU(GN_r(residual))], the arithmetic of
// gn_apply_kernel.  The convolution is bound by the matrix pipe and leaves most of the HBM bandwidth unused; the pass
// is pure streaming.  Unit u of the job belongs to step (tile, chunk) of wave w: u = ((tile * nsteps + chunk) * 8 + w)
// * RIDE + j -- a tile's steps cover 64 KB of the job, the size of the tile's own output.  A unit is loaded at the
// start of a step (two 16-byte loads per lane) and normalised / stored at the start of the next one.
struct RideArgs {
  const float *x, *stats, *gamma, *beta;
  const float *res, *r_stats, *r_gamma, *r_beta;   // res: optional; r_stats: the residual is itself a raw conv output
  float *out;
  int units;        // 256-float units of the job (spatial % 256 == 0: a unit lies inside one (sample, channel) plane)
  WinoDiv fd_upp;   // units per plane
};
typedef const __attribute__((address_space(4))) float *wn_cfloat;   // uniform reads through the scalar cache
__device__ __forceinline__ float wn_sload(const float *p, size_t i) { return ((wn_cfloat)(size_t)p)[i]; }

// U = G g G^T per (cout, cin), packed [chunk of 4 cin][xi = 4i + j][cout tile][lane]; lane = k*16 + c holds
// U_xi[cin = chunk*4 + k][cout = t*16 + c] (the B fragment of the MFMA), zero outside (c_in, 32).
// Volume form (kd = 3, weight (cout, cin, 3, 3, 3)): the chunks of depth tap kz follow those of kz - 1
// (chunk = kz * cin/4 + cin chunk), each the 2-D transform of the tap's 3 x 3 slice.
__global__ void wino_pack_kernel(const float *__restrict__ w, int cin, int cout, int nchunks, int kd,
                                 float *__restrict__ out) {
  const int total = nchunks * WN_UFLOATS;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, t = (idx >> 6) & 1, xi = (idx >> 7) & 15, chunk = idx >> 11;
  const int per_kz = nchunks / kd, kz = chunk / per_kz;
  const int co = t * 16 + (lane & 15), ci = (chunk - kz * per_kz) * 4 + (lane >> 4);
  float u = 0.0f;
  if (co < cout && ci < cin) {
    const float *g = w + (((size_t)co * cin + ci) * kd + kz) * 9;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int i = xi >> 2, j = xi & 3;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) u += G[i][a] * g[a * 3 + b] * G[j][b];
  }
  out[idx] = u;
}

// KS      MFMA k-steps (4 input channels each) per step: 2 for the 32-channel layers (half the barriers per
//         tile), 1 for the 4-channel head
// NSTAGE  depth of the raw-tile ring (KS = 2: 4 x 23 KB next to the 64 KB of U; a step is ~2 us, a DMA round
//         trip under load longer)
// DIL     dilation (1, 2, 4, 8)
// VOL     volume form: 3 x 3 x 3 layer on (n, 32, D, H, W) as the sum over the depth tap kz of the 2-D Winograd
//         products of plane z + kz - 1 (12 multiplies per output instead of 27).  A work item is (sample, plane,
//         tile); its 3 x 4 steps accumulate into one set of registers.  The 192 KB of transformed weights do not
//         fit next to the raw ring, so each step's 16 KB of U travel through a second ring one step behind the
//         raw tiles (the slot of step s is free once every wave has passed the barrier of step s + 1).
// RIDE    256-float units of a carried normalise / activate / add job per wave and step (0: none), see RideArgs
// WIDE    (volume form) tile = 10 rows x 40 columns instead of 16 x 32: planes 32 < W <= 40 columns wide -- the 30x40
//         coarse grid of 640x480 frames -- are 2 x 2 tiles of 16 x 32 at 58.6 % utilisation (the regulariser of BASELINE
//         config 4 ran at 0.40 of the matrix pipe where the 16x32 / 32x64 grids reach 0.68); as 10-row strips of the whole
//         width a plane is 3 tiles of 5 x 20 = 100 patches: 78 %.  The 128 patch slots of a tile are dealt row-major
//         (slot q = 16 wave + lane & 15 -> patch row q / 20, column q % 20; slots >= 100 idle): only the lane -> patch
//         map, the tile constants and the epilogue's per-lane validity differ.
// WIDE 2  ROLLING strips (round 6): the patch rows of a sample's planes are numbered through, R = z * PR + patch row
//         (PR = (H + 1) / 2), and a work item is the SIX consecutive rows R = 6 t .. 6 t + 5 x the whole width = 120 of the
//         128 slots -- wherever they fall: s of them at the bottom of plane zA, the other 6 - s at the top of plane
//         zA + 1.  A 30 x 40 plane (PR = 15) costs 2.5 items instead of 3 (94 % of the slots instead of 78 %).  The raw
//         tile has 15 rows: rows yA0 - 1 .. of plane zA + kz - 1 for the A part, ONE zero row that is both the A part's
//         row H [+ 1] and the B part's row -1, then rows 0 .. of plane zA + kz; both parts of a step come from one
//         descriptor that starts at plane zA + kz - 1 (planes are contiguous), a B lane's offset being one plane further.
//         Per item only s changes: the lane's first raw row (ya: one row further down for B patches), the DMA plan
//         and the epilogue's per-lane plane / row.  Items never span samples (a sample's last item may be short).
constexpr int WN_WIDE_TY = 10, WN_WIDE_TX = 40, WN_WIDE_PC = WN_WIDE_TX / 2, WN_WIDE_NP = (WN_WIDE_TY / 2) * WN_WIDE_PC;
constexpr int WN_ROLL_PR = 6, WN_ROLL_NP = WN_ROLL_PR * WN_WIDE_PC, WN_ROLL_HY = 2 * WN_ROLL_PR + 3;
// FULLW   (round 6) the tile spans the WHOLE plane width (W = 32: the 16 x 32 coarse grid's regulariser): neighbouring
//         patches overlap by two columns, so lane p forms the column sums t[.][1], t[.][2] of ITS OWN two image columns
//         2p, 2p + 1 only (8 adds instead of 16, half the raw-tile reads: one 8-byte read per row) and takes t[.][0] /
//         t[.][3] -- columns 2p - 1 / 2p + 2 -- from lanes p - 1 / p + 1 as DPP operands (row_shr:1 / row_shl:1) of the second
//         stage's adds: 24 VALU instructions per patch and channel instead of 32, no extra instruction.  The edge lanes'
//         missing neighbours are columns -1 and 32, i.e. the zero padding -- exactly what bound_ctrl:0 supplies -- which
//         is why the form needs the tile to span the plane.  Same operands, same operations: bit-identical results.
#ifndef MVSN_WN_NO_FULLW
#define MVSN_WN_FULLW_OK 1
#else
#define MVSN_WN_FULLW_OK 0
#endif
template <int MODE, int KS, int NSTAGE, int DIL, bool VOL = false, int RIDE = 0, int WIDE = 0, bool FULLW = false>
__global__ __launch_bounds__(WN_THREADS, 2) void conv_wino_kernel(WinoArgs g, const float *__restrict__ in,
                                                                  const float *__restrict__ upk,
                                                                  const float *__restrict__ bias,
                                                                  const float *__restrict__ in_stats,
                                                                  const float *__restrict__ in_gamma,
                                                                  const float *__restrict__ in_beta,
                                                                  float *__restrict__ out,
                                                                  float *__restrict__ out_partials, RideArgs rd,
                                                                  MVSN_VIS10) {   // (MVSN_VIS10: mvsn_common.h)
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef MVSN_WN_FORCE_DMA   // tuning aid: 0 builtin everywhere, 1 inline assembly everywhere
  constexpr bool ASM_DMA = MVSN_WN_FORCE_DMA;
#else
  constexpr bool ASM_DMA = RIDE > 0 && DIL == 1;   // see wn_dma16
#endif
#ifdef MVSN_WN_FORCE_BARRIER
  constexpr bool LDS_BARRIER = MVSN_WN_FORCE_BARRIER;
#else
  constexpr bool LDS_BARRIER = !VOL;               // see wn_barrier (r4: with the LDS-only barrier the volume form carrying a
                                                   // pass is no faster either: 14.6 vs 14.4 ms per step for the six carriers)
#endif
  static_assert(!WIDE || (VOL && DIL == 1 && MVSN_WN_TRANSPOSED), "wide tiles: volume form only (transposed epilogue)");
  constexpr bool ROLL = WIDE == 2;
  constexpr int TY = ROLL ? 2 * WN_ROLL_PR : (WIDE ? WN_WIDE_TY : WN_TY), TX = WIDE ? WN_WIDE_TX : WN_TX;
  constexpr int HY = ROLL ? WN_ROLL_HY : TY + 2 * DIL;   // raw rows (ROLL: two parts around one shared zero row)
  constexpr int NP = ROLL ? WN_ROLL_NP : WN_WIDE_NP;     // WIDE: patches among the tile's 128 slots
  constexpr int PA = wn_pa(DIL), XS = TX + 2 * PA, DQ = XS / 4, GROUPS = HY * DQ, PIECES = (GROUPS + 63) / 64;
  constexpr int RCST = HY * XS + 16;
  static_assert(WIDE || (XS == wn_xs(DIL) && GROUPS == wn_groups(DIL) && PIECES == wn_pieces(DIL) && RCST == wn_rcst(DIL)), "");
  constexpr int STAGE = KS * 4 * RCST;               // ring stage (floats)
  float *U = smem + NSTAGE * STAGE;                  // nchunks * WN_UFLOATS, resident (VOL: ring of NSTAGE steps)
  // dilation 1: raw tiles stored one float further (4-byte-aligned DMA destination), see tr_load
  // (FULLW: no shift -- a lane reads its OWN two columns, which then start at an even float)
  static_assert(!FULLW || (DIL == 1 && WIDE == 0 && MVSN_WN_XF == 2 && MVSN_WN_TRANSPOSED), "full-width form: 16 x 32 tiles, burst + block");
  constexpr int SHIFT = (DIL == 1 && MVSN_WN_SHIFT && !FULLW) ? 1 : 0;
  // dilated layers: a half-wave reads two channels whose strided columns fall on the same half of the banks;
  // odd channels are stored DIL floats further, which moves them to the other half (conflict-free)
  constexpr int CSHIFT = (DIL > 1 && MVSN_WN_SHIFT) ? DIL : 0;
  constexpr int UST = KS * WN_UFLOATS;               // U of one step (floats)
  // dilation 1: accumulators as couts x patches (transposed MFMA operand order) -- fully coalesced output stores
  constexpr bool TR = DIL == 1 && MVSN_WN_TRANSPOSED;
  static_assert(!VOL || (KS == 2 && DIL == 1), "volume form: 32 channels in steps of 8, dilation 1");
  const int nsteps = (g.nchunks + KS - 1) / KS;      // steps per tile
  const int uchunks = nsteps * KS;                   // chunks of U in LDS: whole steps (an odd count gets a zero chunk)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t plane = (size_t)g.H * g.W;
  const int ptiles = ROLL ? g.tiles : g.D * g.tiles; // work items per sample (VOL: planes x tiles; ROLL: rolling strips)
  // ROLL: item t of a sample -> plane zA of its first patch row, that row prA0 inside the plane, rows s taken from zA
  auto roll_item = [&](int t, int &zA, int &prA0, int &sA) {
    const int R0 = t * WN_ROLL_PR;
    zA = wdiv(R0, g.fd_pr);
    prA0 = R0 - zA * g.pr;
    sA = g.pr - prA0 < WN_ROLL_PR ? g.pr - prA0 : WN_ROLL_PR;
  };
  const int total = g.n * ptiles;                    // work items in (image, [plane,] tile) order
  const int G = gridDim.x;
  // item of this workgroup in round r: r * G + an XCD-contiguous slot (neighbouring tiles share halo lines in one L2)
  const int slot = xcd_tile_index(blockIdx.x, G);
#ifdef MVSN_WN_STAMPS
  unsigned long long *dbg = (blockIdx.x == gridDim.x / 3 && tid == 0) ? g_wn_stamps : nullptr;
  unsigned long long *dbg_lds = reinterpret_cast<unsigned long long *>(
      U + (VOL ? NSTAGE * UST : uchunks * WN_UFLOATS) + 32 + (RIDE > 0 ? WN_WAVES * RIDE * 256 : 0));
  int dbg_i = 0;
  bool dbg_on = true;
#endif
  WN_STAMP();   // entry

  // ---- resident U: wave w fetches 1 KB runs w, w + 8, ...
  if constexpr (!VOL) {
    const int runs = g.nchunks * (WN_UFLOATS / 256);
    for (int run = wave; run < runs; run += WN_WAVES)
      wn_dma16<ASM_DMA>(upk + (size_t)run * 256 + lane * 4, U + run * 256);
    if (uchunks != g.nchunks)   // (uniform) the zero chunk: 2048 floats, one 16-byte store per thread
      *reinterpret_cast<floatx4 *>(U + g.nchunks * WN_UFLOATS + tid * 4) = floatx4{0.f, 0.f, 0.f, 0.f};
  }

  // the bias is read from LDS in the tile epilogue: as a global load its s_waitcnt vmcnt(0) would drain the DMA ring
  // (which runs ahead into the next tile) once per tile
  float *bias_lds = U + (VOL ? NSTAGE * UST : uchunks * WN_UFLOATS);
  if (tid < 32) bias_lds[tid] = bias ? bias[tid] : 0.0f;   // published by the first barrier

  // ---- prefetcher state: DMA of (item, chunk) steps runs two steps ahead of the multiplies
  // the step's KS * 4 * PIECES pieces are dealt to the 8 waves in order (channel-major); where they do not
  // divide evenly (the 4-channel head: 12 pieces) waves 0-3 take pieces 0, 1 of channel w, waves 4-7 piece 2
  constexpr int STEP_PIECES = KS * 4 * PIECES;
  constexpr bool EVEN = STEP_PIECES % WN_WAVES == 0;
  constexpr int PER = EVEN ? STEP_PIECES / WN_WAVES : 2;   // pieces per wave (upper bound)
  static_assert(EVEN || (KS == 1 && PIECES == 3), "uneven DMA split only for the 12-piece case");
  static_assert(!EVEN || PIECES % PER == 0 || PER % PIECES == 0, "a wave's pieces stay within one channel");
  const int dch = EVEN ? (wave * PER) / PIECES : (wave & 3);
  const int dp0 = EVEN ? (wave * PER) % PIECES : (wave < 4 ? 0 : 2);
  const int dpn = EVEN ? (PER < PIECES ? PER : PIECES) : (wave < 4 ? 2 : 1);
  const float *zero = reinterpret_cast<const float *>(&g_wn_zero16);
  int pf_round = 0, pf_chunk = 0, pf_stage = 0;
  int pf_goff[PER];
  int pf_n = 0, pf_z = 0;
  bool pf_live = slot < total;
  int rd_young = 0;   // RIDE: DMA pieces this wave has issued since its carried loads
  auto pf_plan = [&]() {   // DMA plan of the prefetcher's current item
    const int flat = g.rev ? total - 1 - (pf_round * G + slot) : pf_round * G + slot;
    const int n = flat / ptiles;
    int tile = flat - n * ptiles;
    int r_pr0 = 0, r_s = WN_ROLL_PR;
    if constexpr (ROLL) {
      roll_item(tile, pf_z, r_pr0, r_s);
      tile = 0;
    } else if constexpr (VOL) {
      pf_z = tile / g.tiles;
      tile -= pf_z * g.tiles;
    }
    const int tyi = ROLL ? 0 : wdiv(tile, g.fd_ntx), txi = tile - tyi * g.ntx;
    const int y0 = ROLL ? 2 * r_pr0 : tyi * TY, x0 = txi * TX;
    pf_n = n;
    // (the pieces' rows / columns are re-derived per tile from an opaque copy of the lane id: hoisted out of the tile
    // loop they occupy six registers for the whole launch -- spilled, and reloaded per step, in the carrying kernels)
    int lo;   // (from the hardware, not from `lane`: that one gets spilled around the loops, and a reload here waits vmcnt(0))
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = (dp0 + i) * 64 + lo;
      const int row = e / DQ, q = e - row * DQ;
      int gy = y0 - DIL + row;
      const int gx = x0 - PA + 4 * q;
      int zsh = 0;
      if constexpr (ROLL) {   // raw rows 0 .. 2 s: plane zA from its row y0 - 1; row 2 s + 1: zero; then plane zA + 1 from row 0
        // (selects, no branches; a B piece's offset is one plane further: the step's descriptor starts at the A part's plane)
        const bool isb = (r_s < WN_ROLL_PR) & (row >= 2 * r_s + 1);
        gy = isb ? row - (2 * r_s + 2) : gy;
        zsh = isb ? g.H * g.W : 0;
      }
      const bool ok = (i < dpn) & (e < GROUPS) & (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W);
      pf_goff[i] = ok ? (zsh + gy * g.W + gx) * (MVSN_WN_BUFDMA ? 4 : 1) : -1;
    }
  };
  auto pf_issue = [&]() {   // issue the DMA of the prefetcher's step and advance it
    if (!pf_live) return;
    rd_young += PER;
    bool cok;
    const float *src;
    unsigned rbytes = 0;   // ROLL: bytes the step's descriptor covers
    bool roll_first = false;   // ROLL: the A part's plane is -1 (the B part's plane 0 is where the descriptor starts)
    if constexpr (ROLL) {   // A part: plane zz = zA + kz - 1, B part: plane zz + 1; one descriptor from the A part's plane
      const int kz = pf_chunk >> 2, c = (pf_chunk & 3) * 8 + dch, zz = pf_z + kz - 1;
      cok = zz >= 0 && zz < g.D;
      roll_first = zz < 0;
      // descriptor: from plane max(zz, 0) over the planes of {zz, zz + 1} that exist (min / max arithmetic, no selects:
      // the compiler turned the nested selects into vector code and branches between the multiplies)
      const int zlo = zz < 0 ? 0 : zz, zhi = zz + 1 < g.D ? zz + 1 : g.D - 1;
      const int np = zhi - zlo + 1 > 0 ? zhi - zlo + 1 : 0;        // 2; 1 at the volume's two ends; 0 beyond
      const int zsel = zlo < g.D - 1 ? zlo : g.D - 1;
      // (pinned to scalar registers: left to itself the compiler forms index and size in vector registers and then wraps
      // every DMA piece in a readfirstlane waterfall loop -- three loops per step between the multiplies, +10 % per tile)
      src = in + (size_t)(unsigned)__builtin_amdgcn_readfirstlane((pf_n * 32 + c) * g.D + zsel) * plane;
      rbytes = (unsigned)np * (unsigned)plane * 4u;
      rbytes = (unsigned)__builtin_amdgcn_readfirstlane((int)rbytes);
    } else if constexpr (VOL) {   // step = (depth tap, 8 channels): plane pf_z + kz - 1 of channel c, zeros outside the volume
      const int kz = pf_chunk >> 2, c = (pf_chunk & 3) * 8 + dch, zz = pf_z + kz - 1;
      cok = zz >= 0 && zz < g.D;
      src = in + (((size_t)pf_n * 32 + c) * g.D + (cok ? zz : 0)) * plane;
    } else {
      const int c = pf_chunk * (KS * 4) + dch;
      cok = c < g.cin;
      const int cc = cok ? c : 0, c1 = cc - g.cb0, c2 = c1 - g.cb1;   // wave-uniform: the block select is scalar work
      src = c1 < 0 ? in + (size_t)pf_n * g.bs0 + (size_t)cc * g.cs0
          : c2 < 0 ? g.in1 + ((size_t)pf_n * g.cb1 + c1) * plane
                   : g.in2 + ((size_t)pf_n * (g.cin - g.cb0 - g.cb1) + c2) * plane;
    }
    float *dst = smem + pf_stage * STAGE + dch * RCST + SHIFT + CSHIFT * (dch & 1);
    const unsigned pbytes = ROLL ? rbytes : (cok ? (unsigned)plane * 4u : 0u);   // (uniform) an empty descriptor: zeros for every lane
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (i < dpn) {   // uniform
        if ((dp0 + i) * 64 + lane < GROUPS) {   // lanes past the tile's last 16-byte group stay out of the slot
          if constexpr (ROLL) {
            static_assert(!ROLL || MVSN_WN_BUFDMA, "rolling strips: descriptor DMA only");
            unsigned off = (unsigned)pf_goff[i];
            // (uniform, one step in D x 12: the A part's plane is -1 -- the descriptor starts at plane 0, which is the B
            // part's: B offsets come back by one plane, A offsets wrap far out of range)
            off -= roll_first ? (unsigned)plane * 4u : 0u;
            wn_dma16_buf<ASM_DMA>(src, pbytes, off, dst + (dp0 + i) * 256);
          } else if constexpr (MVSN_WN_BUFDMA) {
            wn_dma16_buf<ASM_DMA>(src, pbytes, (unsigned)pf_goff[i], dst + (dp0 + i) * 256);   // byte offset; -1: out of range
          } else {
            const float *p = (cok && pf_goff[i] >= 0) ? src + pf_goff[i] : zero;
            wn_dma16<ASM_DMA>(p, dst + (dp0 + i) * 256);
          }
        }
      }
    }
    pf_stage = pf_stage + 1 == NSTAGE ? 0 : pf_stage + 1;
    if (++pf_chunk == nsteps) {
      pf_chunk = 0;
      ++pf_round;
      pf_live = pf_round * G + slot < total;
      if (pf_live) pf_plan();
    }
  };
  // VOL: the U of the steps, two 1 KB pieces per wave, one step behind the raw ring
  int uq_chunk = 0, uq_stage = 0, uq_left = 0;
  auto uq_issue = [&]() {
    if constexpr (VOL) {
      if (uq_left <= 0) return;
      --uq_left;
      rd_young += 2;
      const float *src = upk + (size_t)uq_chunk * UST + wave * 512 + lane * 4;
      float *dst = U + uq_stage * UST + wave * 512;
      wn_dma16<ASM_DMA>(src, dst);
      wn_dma16<ASM_DMA>(src + 256, dst + 256);
      uq_stage = uq_stage + 1 == NSTAGE ? 0 : uq_stage + 1;
      uq_chunk = uq_chunk + 1 == nsteps ? 0 : uq_chunk + 1;
    }
  };
  if (pf_live) pf_plan();
  if constexpr (VOL) uq_left = slot < total ? ((total - slot + G - 1) / G) * nsteps : 0;
#pragma unroll
  for (int i = 0; i < NSTAGE; ++i) {
    pf_issue();
    if (i < NSTAGE - 1) uq_issue();
  }
  WN_STAMP();   // prologue

  const int pcol = lane & 15, kc = lane >> 4;   // this lane's patch column / channel within the chunk; patch row = wave
  // first output row / column of patch row `wave` / patch column `pcol` inside the tile (the second is + DIL)
  int ya = (wave / DIL) * 2 * DIL + wave % DIL, xa = (pcol / DIL) * 2 * DIL + pcol % DIL;
  int lane_pr = 0;        // WIDE: this lane's patch row inside the tile
  if constexpr (WIDE != 0) {   // slot q of the tile -> patch (q / 20, q % 20); idle slots read patch 0 (their outputs are never stored)
    const int q = wave * 16 + pcol, qq = q < NP ? q : 0;
    const int pr = qq / WN_WIDE_PC;
    lane_pr = pr;
    ya = 2 * pr, xa = 2 * (qq - pr * WN_WIDE_PC);   // (ROLL: ya is set per item by tr_setup -- B patches sit one raw row further down)
  }
  const int my_items = slot < total ? (total - slot + G - 1) / G : 0;
  const int total_steps = my_items * nsteps;

  // wait until this wave's DMA pieces of a step have landed, `younger` later steps having been issued since
  // (vmcnt retires in order; waves 0-3 issue two pieces per step, waves 4-7 one)
  constexpr int PERV = PER + (VOL ? 2 : 0);   // + the step's two U pieces
  static_assert(!VOL || EVEN, "volume form: every wave issues the same number of pieces");
  static_assert(RIDE == 0 || EVEN, "carried jobs: layers with an even DMA split");
  auto wait_landed = [&](int younger) {
#define WN_WAIT_CASE(K)                                                              \
  case K:                                                                            \
    if (EVEN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PERV * (K)) : "memory");        \
    else if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (K)) : "memory"); \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");                       \
    break;
    switch (younger) {   // uniform
      WN_WAIT_CASE(0)
      WN_WAIT_CASE(1)
      WN_WAIT_CASE(2)
      WN_WAIT_CASE(3)
      WN_WAIT_CASE(4)
      default:
        WN_WAIT_CASE(5)
    }
#undef WN_WAIT_CASE
  };
  static_assert(NSTAGE - 1 <= 5 && PERV * 5 <= 63, "wait_landed covers up to 5 younger steps within the vmcnt range");

  // ---- MODE 1: the previous layer's LeakyReLU(GroupNorm(.)) is applied IN LDS by the wave that fetched a piece,
  // once per element, between the piece's arrival and the barrier that publishes the step (the patches overlap
  // 2 x 2: transformed in registers per patch every element would be processed four times).  Pieces outside
  // the image keep their zeros, as the padding of the materialised tensor would be.
  int xf_round = 0, xf_chunk = 0, xf_stage = 0;
  unsigned xf_mask = 0;            // which of this wave's pieces of the current tile lie inside the image
  unsigned xf_maska = 0, xf_maskb = 0;   // ROLL: ... of the A part / of the B part (xf_mask = those whose plane exists this step)
  float xf_sc = 0.f, xf_sh = 0.f;  // scale / shift of this wave's channel of the current step
  bool xf_on = false;
  auto xf_prepare = [&]() {        // parameters of the step the in-LDS side handles next (issued one step early)
    if constexpr (MODE == 1) {
      const int lin = xf_round * G + slot;
      const int flat = g.rev ? total - 1 - lin : lin;
      xf_on = lin < total;
      if (!xf_on) return;
      const int n = flat / ptiles;
      int tile = flat - n * ptiles, z = 0;
      int r_pr0 = 0, r_s = WN_ROLL_PR;
      if constexpr (ROLL) {
        roll_item(tile, z, r_pr0, r_s);
        tile = 0;
      } else if constexpr (VOL) {
        z = tile / g.tiles;
        tile -= z * g.tiles;
      }
      if (xf_chunk == 0) {
        const int tyi = ROLL ? 0 : wdiv(tile, g.fd_ntx), txi = tile - tyi * g.ntx;
        const int y0 = ROLL ? 2 * r_pr0 : tyi * TY, x0 = txi * TX;
        xf_mask = 0;
        xf_maska = xf_maskb = 0;
        int lo;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int e = (dp0 + i) * 64 + lo;
          const int row = e / DQ, q = e - row * DQ;
          int gy = y0 - DIL + row;
          const int gx = x0 - PA + 4 * q;
          bool isb = false;
          if constexpr (ROLL) {
            isb = r_s < WN_ROLL_PR && row >= 2 * r_s + 1;
            if (isb) gy = row - (2 * r_s + 2);
          }
          if (i < dpn && e < GROUPS && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
            if constexpr (ROLL) (isb ? xf_maskb : xf_maska) |= 1u << i;
            else xf_mask |= 1u << i;
          }
        }
      }
      int c = xf_chunk * (KS * 4) + dch;   // wave-uniform: scalar loads
      bool live = c < g.cin;
      if constexpr (VOL) {   // planes outside the volume keep their zeros
        const int zz = z + (xf_chunk >> 2) - 1;
        c = (xf_chunk & 3) * 8 + dch;
        live = zz >= 0 && zz < g.D;
        if constexpr (ROLL) {   // per part: the A part's plane zz, the B part's zz + 1
          const bool liveb = zz + 1 < g.D;
          xf_mask = (live ? xf_maska : 0u) | (liveb ? xf_maskb : 0u);
          live = live || liveb;
        }
      }
      if (live) {
        const float mean = in_stats[((size_t)n * 4 + (c >> 3)) * 2 + 0];
        const float rstd = in_stats[((size_t)n * 4 + (c >> 3)) * 2 + 1];
        xf_sc = rstd * in_gamma[c];
        xf_sh = in_beta[c] - mean * xf_sc;
      } else {
        xf_on = false;
      }
    }
  };
  auto xf_apply = [&]() {          // ... applied to this wave's landed pieces; then the side moves on one step
    if constexpr (MODE == 1) {
      if (xf_on) {
        float *dst = smem + xf_stage * STAGE + dch * RCST + lane * 4 + SHIFT + CSHIFT * (dch & 1);
#pragma unroll
        for (int i = 0; i < PER; ++i)
          if ((xf_mask >> i) & 1u) {
            float *q = dst + (dp0 + i) * 256;
            if constexpr (SHIFT || CSHIFT % 4 != 0) {   // the shifted tile is only 4- / 8-byte aligned: dword pairs
              float e[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
              for (int r = 0; r < 4; ++r) e[r] = lrelu02(e[r] * xf_sc + xf_sh);
              q[0] = e[0], q[1] = e[1], q[2] = e[2], q[3] = e[3];
            } else {
              floatx4 v = *reinterpret_cast<floatx4 *>(q);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = lrelu02(v[r] * xf_sc + xf_sh);
              *reinterpret_cast<floatx4 *>(q) = v;
            }
          }
      }
      xf_stage = xf_stage + 1 == NSTAGE ? 0 : xf_stage + 1;
      if (++xf_chunk == nsteps) xf_chunk = 0, ++xf_round;
    }
  };

  // ---- transform side: runs one step ahead of the multiplies (its tile may already be the next one)
  int tr_round = 0, tr_chunk = 0, tr_stage = 0;
  // (nothing per tile on this side -- MODE 1 is applied in LDS by the fetching wave -- except ROLL: the lane's first raw
  // row depends on how the item splits)
  auto tr_setup = [&]() {
    if constexpr (ROLL) {
      const int lin = tr_round * G + slot;
      if (lin < total) {
        const int flat = g.rev ? total - 1 - lin : lin;
        int zA, prA0, sA;
        roll_item(flat - (flat / ptiles) * ptiles, zA, prA0, sA);
        ya = 2 * lane_pr + (lane_pr >= sA ? 1 : 0);
      }
    }
  };
  // the 4 x 4 patch of (channel kc, patch pcol of patch row wave) of the transform side's step, transformed:
  // the result IS the A fragment of the 16 coefficient GEMMs
  // raw patch as loaded, d[h][row][slot]:
  //   SHIFT (dilation 1): the tile sits one float further in LDS, so a patch starts at an even column and a row is
  //   two aligned 8-byte reads (one ds_read2_b64, all 32 banks busy); slot = column;
  //   otherwise: slots (0, 1) = columns (0, 2), slots (2, 3) = columns (1, 3) -- the register pairs two
  //   ds_read2_b32 produce, consumed in place (no re-interleaving moves).
  auto dslot = [](int j) { return SHIFT ? j : (j & 1) * 2 + (j >> 1); };
  auto tr_load = [&](float (&d)[KS][4][4], int h0 = 0, int h1 = KS) {
#pragma unroll
    for (int h = 0; h < KS; ++h) {
      if (h < h0 || h >= h1) continue;
      const float *raw = smem + tr_stage * STAGE + (h * 4 + kc) * RCST + ya * XS + xa + (PA - DIL) + SHIFT + CSHIFT * (kc & 1);
      if constexpr (FULLW) {   // own columns j = 1, 2 of the patch (image columns 2p, 2p + 1): one aligned 8-byte read per row
        const float2 *r2 = reinterpret_cast<const float2 *>(raw + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 c = r2[i * (XS / 2)];
          d[h][i][1] = c.x, d[h][i][2] = c.y;
        }
      } else if constexpr (SHIFT) {
        const float2 *r2 = reinterpret_cast<const float2 *>(raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 lo = r2[i * (XS / 2)], hi = r2[i * (XS / 2) + 1];
          d[h][i][0] = lo.x, d[h][i][1] = lo.y, d[h][i][2] = hi.x, d[h][i][3] = hi.y;
        }
      } else if constexpr (DIL == 1) {
        // columns (0, 2) and (1, 3) from two bases the compiler cannot relate: it would otherwise fuse the
        // middle pair into a ds_read_b64 (three LDS instructions per row instead of two ds_read2_b32)
        int oa = 0, ob = 1;
        asm volatile("" : "+v"(oa), "+v"(ob));
        const float *ra = raw + oa, *rb = raw + ob;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[h][i][0] = ra[i * XS], d[h][i][1] = ra[i * XS + 2];
          d[h][i][2] = rb[i * XS], d[h][i][3] = rb[i * XS + 2];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[h][i][0] = raw[i * DIL * XS], d[h][i][1] = raw[i * DIL * XS + 2 * DIL];
          d[h][i][2] = raw[i * DIL * XS + DIL], d[h][i][3] = raw[i * DIL * XS + 3 * DIL];
        }
      }
    }
  };
  auto tr_finish = [&](float (&d)[KS][4][4], float (&v)[KS][16], int h0 = 0, int h1 = KS) {
#pragma unroll
    for (int h = 0; h < KS; ++h) {
      if (h < h0 || h >= h1) continue;
      if constexpr (FULLW) {
        // column sums of the lane's own columns; the outer two from the neighbouring patches (lanes p -+ 1 of the 16-lane
        // row: DPP operands of the adds below; lanes 0 / 15 get 0 = the padding columns -1 / 32)
        float t1[4], t2[4];
        {
          const float a0 = d[h][0][1], a1 = d[h][1][1], a2 = d[h][2][1], a3 = d[h][3][1];
          const float b0 = d[h][0][2], b1 = d[h][1][2], b2 = d[h][2][2], b3 = d[h][3][2];
          t1[0] = a0 - a2, t1[1] = a1 + a2, t1[2] = a2 - a1, t1[3] = a1 - a3;
          t2[0] = b0 - b2, t2[1] = b1 + b2, t2[2] = b2 - b1, t2[3] = b1 - b3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[h][i * 4 + 0] = dpp_mov<0x111>(t2[i]) - t2[i];   // t[i][0] = lane p - 1's t[i][2]   (row_shr:1)
          v[h][i * 4 + 1] = t1[i] + t2[i];
          v[h][i * 4 + 2] = t2[i] - t1[i];
          v[h][i * 4 + 3] = t1[i] - dpp_mov<0x101>(t1[i]);   // t[i][3] = lane p + 1's t[i][1]   (row_shl:1)
        }
        continue;
      }
      float t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d0 = d[h][0][dslot(j)], d1 = d[h][1][dslot(j)], d2 = d[h][2][dslot(j)], d3 = d[h][3][dslot(j)];
        t[0][j] = d0 - d2;
        t[1][j] = d1 + d2;
        t[2][j] = d2 - d1;
        t[3][j] = d1 - d3;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[h][i * 4 + 0] = t[i][0] - t[i][2];
        v[h][i * 4 + 1] = t[i][1] + t[i][2];
        v[h][i * 4 + 2] = t[i][2] - t[i][1];
        v[h][i * 4 + 3] = t[i][1] - t[i][3];
      }
    }
  };
  auto tr_advance = [&]() {   // the transform side moves on one step
    tr_stage = tr_stage + 1 == NSTAGE ? 0 : tr_stage + 1;
    if (++tr_chunk == nsteps) {
      tr_chunk = 0;
      ++tr_round;
      tr_setup();
    }
  };

  // ---- RIDE: the carried job.  A step's RIDE consecutive units (one plane: units per plane % RIDE == 0) are fetched
  // during it -- x into registers, the residual by LDS-DMA into the wave's own 1 KB slots behind U (no registers
  // held across the multiplies, no other wave reads them: no barrier) -- and normalised / stored at the start of
  // the next step: the only entries of the (in-order) vmcnt queue younger than them are that step's DMA pieces.
  // Every step fetches and every step consumes (steps without a unit re-read unit 0 and store nothing): with the
  // use conditional the compiler has to assume loads pending on the other path and drains the queue before it
  // reuses their registers.
  constexpr int RN = RIDE > 0 ? RIDE : 1;
  floatx4 rd_v[RN];
  float rd_sc = 0.f, rd_sh = 0.f, rd_rsc = 0.f, rd_rsh = 0.f;   // wave-uniform
  int rd_u = 0;                                                 // first unit in flight
  bool rd_ok = false;                                           // ... is one of the job's
  float *rds = U + (VOL ? NSTAGE * UST : uchunks * WN_UFLOATS) + 32 + wave * (RN * 256);
  auto rd_issue = [&](int flat, int chunk) {   // flat < 0: nothing to fetch (the set-up's and the last step's)
    if constexpr (RIDE > 0) {
      const int u = ((flat * nsteps + chunk) * WN_WAVES + wave) * RN;
      rd_ok = flat >= 0 && u < rd.units;
      rd_u = rd_ok ? u : 0;
      const int pl = wdiv(rd_u, rd.fd_upp), c = pl & 31, n = pl >> 5;
      const float mean = wn_sload(rd.stats, ((size_t)n * 4 + (c >> 3)) * 2 + 0);
      const float rstd = wn_sload(rd.stats, ((size_t)n * 4 + (c >> 3)) * 2 + 1);
      rd_sc = rstd * wn_sload(rd.gamma, c);
      rd_sh = wn_sload(rd.beta, c) - mean * rd_sc;
      // (descriptors on the step's units: scalar base, the lane's offset is its constant 16 bytes)
      rd_young = 0;
      const __amdgpu_buffer_rsrc_t xs = wn_rsrc(rd.x + (size_t)rd_u * 256, RN * 1024);
#pragma unroll
      for (int j = 0; j < RN; ++j)
        if (!(MVSN_RD_ABLATE & 1))
          rd_v[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xs, lane * 16 + j * 1024, 0, 2));
      if (rd.res && !(MVSN_RD_ABLATE & 4)) {
#pragma unroll
        for (int j = 0; j < RN; ++j)
          wn_dma16_buf<ASM_DMA>(rd.res + (size_t)rd_u * 256, RN * 1024, (unsigned)(lane * 16 + j * 1024), rds + j * 256);
      }
      if (rd.r_stats) {
        const float rm = wn_sload(rd.r_stats, ((size_t)n * 4 + (c >> 3)) * 2 + 0);
        rd_rsc = wn_sload(rd.r_stats, ((size_t)n * 4 + (c >> 3)) * 2 + 1) * wn_sload(rd.r_gamma, c);
        rd_rsh = wn_sload(rd.r_beta, c) - rm * rd_rsc;
      }
    }
  };
  auto rd_consume = [&](int ln) {
    if constexpr (RIDE > 0) {
      // all arithmetic first, then the stores: a store in flight next to a load still awaited makes the compiler
      // drain the queue (it treats mixed loads / stores as unordered) -- the store's whole latency, every step
      floatx4 o[RN];
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        if (MVSN_RD_ABLATE & 16) asm volatile("" ::"v"(rd_v[j]));   // loads kept, nothing done with them
        if (MVSN_RD_ABLATE & 2) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[j][k] = lrelu02(rd_v[j][k] * rd_sc + rd_sh);
        if (rd.res) {
          const floatx4 r = *reinterpret_cast<const floatx4 *>(rds + j * 256 + ln * 4);
          if (rd.r_stats) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[j][k] += lrelu02(r[k] * rd_rsc + rd_rsh);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[j][k] += r[k];
          }
        }
      }
      if (rd_ok && !(MVSN_RD_ABLATE & 2)) {   // uniform
        const __amdgpu_buffer_rsrc_t os = wn_rsrc(rd.out + (size_t)rd_u * 256, RN * 1024);
#pragma unroll
        for (int j = 0; j < RN; ++j) wn_store4<2>(os, (unsigned)(ln * 16 + j * 1024), 0, o[j]);
      }
    }
  };

  float v[KS][16];
  if (total_steps > 0) {
    tr_setup();
    xf_prepare();
    wait_landed(total_steps - 1 < NSTAGE - 1 ? total_steps - 1 : NSTAGE - 1);
    xf_apply();        // step 0
    xf_prepare();      // step 1
    wn_barrier<LDS_BARRIER>();      // step 0 (and U) visible to everyone
    float d0[KS][4][4];
    tr_load(d0);
    tr_finish(d0, v);
    tr_advance();
    rd_issue(-1, 0);
  }

  // ---- a finished tile: output transform, bias, stores, GroupNorm partials
  floatx4 acc[16][2];
  auto finish_tile = [&](int n, int z, int tile_id, int y0, int x0) {
    // lane: cout t*16 + (lane&15); patches p = 4*(lane>>4) + r of patch row `wave`: output rows ya, ya + DIL and
    // columns xa(p), xa(p) + DIL.  Whatever the dilation, the eight columns of a lane's four patches form two
    // aligned groups of four consecutive columns: element (r, second) goes to slot k of group h.
    // The lane id is re-derived here (two instructions the optimiser cannot hoist): everything the epilogue forms
    // from it -- output and record addresses, 64-bit per lane -- would otherwise be computed once in front of the tile
    // loop and carried through it; in the carrying instantiations those values (and `lane` itself) were SPILLED, and a
    // reload in the epilogue comes with s_waitcnt vmcnt(0): a drain of the tile's output stores and of the DMA ring.
    int lq;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lq));
    const int cl = lq & 15, gq = lq >> 4;
    constexpr int PSH = DIL <= 2 ? 1 : 0;   // DIL 1, 2: h = r >> 1; DIL 4, 8: h = second
    auto slot_h = [](int r, int sec) { return PSH ? (r >> 1) : sec; };
    auto slot_k = [](int r, int sec) { return DIL == 1 ? 2 * (r & 1) + sec : (DIL == 2 ? (r & 1) + 2 * sec : r); };
    const int oy = y0 + ya;
    const int xg0 = DIL == 8 ? 16 * (gq >> 1) + 4 * (gq & 1) : 8 * gq;   // first column of group 0 inside the tile
    const int xg1 = xg0 + (DIL == 8 ? 8 : 4);
    const bool row0 = oy < g.H, row1 = oy + DIL < g.H;
    const bool q0 = x0 + xg0 < g.W, q1 = x0 + xg1 < g.W;   // W % 4 == 0: each float4 is all inside or all outside
    const size_t cstride = VOL ? (size_t)g.D * plane : plane;   // output channel stride
    // descriptor of this sample's [plane z of the] 32 output channels; lane part: its cout and first column
    const __amdgpu_buffer_rsrc_t osrd = wn_rsrc(out + (size_t)n * 32 * cstride + (size_t)z * plane, (unsigned)(32 * cstride * 4));
    const unsigned obase = ((unsigned)cl * (unsigned)cstride + (unsigned)(oy * g.W + x0)) * 4u;
    const unsigned ovoff0 = q0 ? obase + (unsigned)xg0 * 4u : 0xFFFFFFFFu, ovoff1 = q1 ? obase + (unsigned)xg1 * 4u : 0xFFFFFFFFu;
    const int cnt = ((row0 ? 1 : 0) + (row1 ? 1 : 0)) * ((q0 ? 4 : 0) + (q1 ? 4 : 0));
    float s[2] = {0.f, 0.f};
    float y[2][2][8];   // [t][row][group * 4 + slot]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float bv = bias_lds[t * 16 + cl];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0[j] = acc[j][t][r] + acc[4 + j][t][r] + acc[8 + j][t][r];
          s1[j] = acc[4 + j][t][r] - acc[8 + j][t][r] - acc[12 + j][t][r];
        }
        y[t][0][slot_h(r, 0) * 4 + slot_k(r, 0)] = s0[0] + s0[1] + s0[2] + bv;
        y[t][0][slot_h(r, 1) * 4 + slot_k(r, 1)] = s0[1] - s0[2] - s0[3] + bv;
        y[t][1][slot_h(r, 0) * 4 + slot_k(r, 0)] = s1[0] + s1[1] + s1[2] + bv;
        y[t][1][slot_h(r, 1) * 4 + slot_k(r, 1)] = s1[1] - s1[2] - s1[3] + bv;
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const bool rok = rr ? row1 : row0;   // uniform
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = rok && (h ? q1 : q0);
          if (rok && (!(MVSN_WN_ABLATE & 64) || n < 0))   // (tuning aid: bit 6 = no output stores)
            wn_store4<MVSN_WN_NT_STORES ? 2 : 0>(
                osrd, h ? ovoff1 : ovoff0, (unsigned)(((size_t)(t * 16) * cstride + (size_t)rr * DIL * g.W) * 4),
                floatx4{y[t][rr][4 * h], y[t][rr][4 * h + 1], y[t][rr][4 * h + 2], y[t][rr][4 * h + 3]});
          if (ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[t] += y[t][rr][4 * h + k];
          }
        }
      }
    }
    WN_STAMP();   // output transform + stores issued
    if (out_partials != nullptr) {   // uniform
      // One record per (wave, 16-lane row): the 8 lanes of a half row hold the 8 couts of one GroupNorm group for
      // the same 16 outputs, so their validity is identical, the block count is a power of two and the whole
      // reduction is three DPP adds (no LDS round trips; cross-row shuffles cost ~10 % of the layer).  The records
      // carry (count, mean, M2) of their 128 values; mvsn_groupnorm_finalize combines them (Chan et al.).
      auto sum8 = [](float v) {
        v += dpp_mov<0xB1>(v);    // quad_perm [1, 0, 3, 2]
        v += dpp_mov<0x4E>(v);    // quad_perm [2, 3, 0, 1]
        v += dpp_mov<0x141>(v);   // row_half_mirror
        return v;
      };
      const int hi = (lq >> 3) & 1;
      const float npos = 8.0f * (float)cnt;
      const float rn = cnt == 16 ? 0.0078125f : (cnt == 8 ? 0.015625f : (cnt == 4 ? 0.03125f : 0.0f));   // 1 / npos, exact
      float m[2], qv[2] = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        s[t] = sum8(s[t]);
        m[t] = s[t] * rn;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if ((rr ? row1 : row0) && (h ? q1 : q0)) {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float dv = y[t][rr][4 * h + k] - m[t];
                qv[t] += dv * dv;
              }
            }
#pragma unroll
      for (int t = 0; t < 2; ++t) qv[t] = sum8(qv[t]);
      if ((lq & 7) == 0) {   // lanes 0 and 8 of every row
        float *rec = out_partials + (((size_t)n * ptiles + (size_t)z * g.tiles + tile_id) * WN_WAVES + wave) * 48;   // uniform
        rec += gq * 12 + hi * 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          rec[t * 6 + 0] = npos;
          rec[t * 6 + 1] = m[t];
          rec[t * 6 + 2] = qv[t];
        }
      }
    }
  };

  // ---- the same for the transposed accumulators (TR, dilation 1): lane = (patch pcol of patch row `wave`, cout quad gq);
  // acc[xi][t][r] = M_xi[cout t*16 + 4 gq + r][patch].  A patch's 2 x 2 outputs are two float2 stores (rows 2 wave,
  // 2 wave + 1; columns 2 pcol, 2 pcol + 1): the 16 lanes of a row write 128 contiguous bytes of one output row of one
  // cout -- four full lines per instruction.  In the patches x couts form a store instruction scattered 64 sixteen-byte
  // pieces over 16 couts and relied on L2 to merge eight partial writes per line: the stores were 11 % of a 2-D layer
  // (ablation, round 4), and as non-temporal stores (no merging) the layer was 10 % SLOWER.
  // GroupNorm records: the 16 lanes of row gq hold, per cout tile t, 4 couts x 4 outputs of group 2 t + (gq >> 1) for
  // their 16 patches: one record per (wave, gq) as before, with data for two of its four groups and count 0 for the
  // others (gn_finalize adds cnt, cnt * mean, M2 + cnt * mean^2: an empty group contributes nothing).
  // (ROLL: z = the A part's plane, y0 = its first output row, r_s = patch rows it holds; tile_id = the item of the sample)
  auto finish_tile_tr = [&](int n, int z, int tile_id, int y0, int x0, int r_s) {
    int lq;   // (re-derived, see finish_tile)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lq));
    const int pc = lq & 15, gq = lq >> 4;
    int oy = y0 + 2 * wave, ox = x0 + 2 * pc;
    bool slot_ok = true;   // WIDE: the tile's 128 slots hold 100 (ROLL: 120) patches
    unsigned zb = 0;       // ROLL: float offset of this lane's plane from plane z (B patches: one plane further)
    if constexpr (WIDE != 0) {
      const int q = wave * 16 + pc;
      slot_ok = q < NP;
      const int qq = slot_ok ? q : 0, pr = qq / WN_WIDE_PC;
      oy = y0 + 2 * pr, ox = x0 + 2 * (qq - pr * WN_WIDE_PC);
      if constexpr (ROLL) {   // a B patch: rows from the top of plane z + 1 (which may not exist: the sample's last item)
        const bool pb = pr >= r_s;
        oy = pb ? 2 * (pr - r_s) : oy;
        zb = pb ? (unsigned)plane : 0u;
        slot_ok = slot_ok & (!pb | (z + 1 < g.D));
      }
    }
    // rows: wave-uniform in the 16 x 32 form, per lane in the WIDE form (there a row past the image is an out-of-range
    // offset like a column past it, and the stores are unconditional)
    const bool row0 = oy < g.H, row1 = oy + 1 < g.H, cok = slot_ok && ox < g.W;   // W % 4 == 0: a column pair is inside or outside
    const size_t cstride = VOL ? (size_t)g.D * plane : plane;
    const __amdgpu_buffer_rsrc_t osrd = wn_rsrc(out + (size_t)n * 32 * cstride + (size_t)z * plane, (unsigned)(32 * cstride * 4));
    const unsigned ovoff = cok ? ((unsigned)(4 * gq) * (unsigned)cstride + zb + (unsigned)(oy * g.W + ox)) * 4u : 0xFFFFFFFFu;
    const unsigned ovoff0 = (!WIDE || row0) ? ovoff : 0xFFFFFFFFu, ovoff1 = (!WIDE || row1) ? ovoff : 0xFFFFFFFFu;
    float s[2] = {0.f, 0.f};
    float y[2][4][4];   // [t][r][2 * row + column]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const floatx4 bv = *reinterpret_cast<const floatx4 *>(bias_lds + t * 16 + 4 * gq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0[j] = acc[j][t][r] + acc[4 + j][t][r] + acc[8 + j][t][r];
          s1[j] = acc[4 + j][t][r] - acc[8 + j][t][r] - acc[12 + j][t][r];
        }
        y[t][r][0] = s0[0] + s0[1] + s0[2] + bv[r];
        y[t][r][1] = s0[1] - s0[2] - s0[3] + bv[r];
        y[t][r][2] = s1[0] + s1[1] + s1[2] + bv[r];
        y[t][r][3] = s1[1] - s1[2] - s1[3] + bv[r];
        if (!(MVSN_WN_ABLATE & 64) || n < 0) {   // (rows: uniform; columns outside the image: dropped by the range check)
          const unsigned so = (unsigned)((size_t)(t * 16 + r) * cstride * 4);
          if (WIDE || row0) wn_store2<MVSN_WN_NT_TR ? 2 : 0>(osrd, ovoff0, so, y[t][r][0], y[t][r][1]);
          if (WIDE || row1) wn_store2<MVSN_WN_NT_TR ? 2 : 0>(osrd, ovoff1, so + (unsigned)g.W * 4u, y[t][r][2], y[t][r][3]);
        }
        if (row0 && cok) s[t] += y[t][r][0] + y[t][r][1];
        if (row1 && cok) s[t] += y[t][r][2] + y[t][r][3];
      }
    }
    WN_STAMP();   // output transform + stores issued
    if (out_partials != nullptr) {   // uniform
      auto sum16 = [](float v) {   // all 16 lanes of a row end up with the row's sum
        v += dpp_mov<0xB1>(v);    // quad_perm [1, 0, 3, 2]
        v += dpp_mov<0x4E>(v);    // quad_perm [2, 3, 0, 1]
        v += dpp_mov<0x141>(v);   // row_half_mirror
        v += dpp_mov<0x140>(v);   // row_mirror
        return v;
      };
      // valid outputs of the record: 4 couts x rows x 2 columns x the tile's valid patch columns (uniform)
      const int vp = (g.W - x0) >> 1;
      float npos = (float)(8 * ((row0 ? 1 : 0) + (row1 ? 1 : 0)) * (vp > 16 ? 16 : vp));
      if constexpr (WIDE != 0)   // per lane: 4 couts x 2 columns x the valid rows of the record's valid patches
        npos = sum16(cok ? (float)(8 * ((row0 ? 1 : 0) + (row1 ? 1 : 0))) : 0.0f);
      float m[2], qv[2] = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        s[t] = sum16(s[t]);
        m[t] = npos > 0.f ? s[t] / npos : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((k < 2 ? row0 : row1) && cok) {
              const float dv = y[t][r][k] - m[t];
              qv[t] += dv * dv;
            }
#pragma unroll
      for (int t = 0; t < 2; ++t) qv[t] = sum16(qv[t]);
      if (pc < 4) {   // lane j of the row writes group j of the record: the row's two groups, zeros for the others
        float *rec = out_partials + (((size_t)n * ptiles + (ROLL ? (size_t)0 : (size_t)z * g.tiles) + tile_id) * WN_WAVES + wave) * 48;   // uniform
        rec += gq * 12 + pc * 3;
        const int ga = gq >> 1;
        const bool a = pc == ga, b = pc == 2 + ga;
        rec[0] = (a || b) ? npos : 0.f;
        rec[1] = a ? m[0] : (b ? m[1] : 0.f);
        rec[2] = a ? qv[0] : (b ? qv[1] : 0.f);
      }
    }
  };

  int step = 0, mm_stage = 0;   // mm_stage: U ring slot of `step` (VOL)
  for (int round = 0; round < my_items; ++round) {
#ifdef MVSN_WN_STAMPS
    dbg_on = round >= 3 || my_items < 4;   // steady state: skip the first tiles (few tiles: the launch's whole life)
#endif
    const int flat = g.rev ? total - 1 - (round * G + slot) : round * G + slot;
    const int n = flat / ptiles;
    int tile_id = flat - n * ptiles, z = 0;
    int r_pr0 = 0, r_s = WN_ROLL_PR;
    if constexpr (ROLL) {
      roll_item(tile_id, z, r_pr0, r_s);
    } else if constexpr (VOL) {
      z = tile_id / g.tiles;
      tile_id -= z * g.tiles;
    }
    const int tyi = ROLL ? 0 : wdiv(tile_id, g.fd_ntx), txi = ROLL ? 0 : tile_id - tyi * g.ntx;
    const int y0 = ROLL ? 2 * r_pr0 : tyi * TY, x0 = txi * TX;

    // acc is first written by the tile's first 32 multiplies (C = 0): no zeroing pass.  The empty asm "defines"
    // the registers here, so the allocator does not carry 128 undefined values around the tile loop.
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) asm volatile("" : "=v"(acc[xi][0]), "=v"(acc[xi][1]));

    // The tile's first step is peeled (its first multiplies run with C = 0): as a branch inside the loop the two
    // multiply variants would meet in a join of 128 accumulators + 32 coefficients, which the allocator spills.
    int chunk = 0;
    auto do_step = [&](auto firstc) {
      const bool has_next = step + 1 < total_steps;   // uniform
      bool waited = false;
      if constexpr (RIDE > 0) {
        // Everything up to the carried loads has landed once only the previous step's DMA pieces are outstanding --
        // which covers step + 1 (issued NSTAGE - 1 >= 2 steps ago) as well.
        if (rd_young == PERV) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PERV) : "memory");
        else if (rd_young == PER) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else if (VOL && rd_young == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rd_consume(lane);
        waited = true;
      }
      if (has_next) {
        const int rest = total_steps - (step + 2);     // steps issued after step + 1 so far
        if (!waited) wait_landed(rest < 0 ? 0 : (rest < NSTAGE - 2 ? rest : NSTAGE - 2));   // step + 1 has landed
        xf_apply();        // step + 1
        WN_STAMP();   // landed
        if (!(MVSN_WN_ABLATE & 8)) wn_barrier<LDS_BARRIER>();   // ... for everyone; everyone has read the raw tile of `step`
        if (MVSN_WN_STAGGER > 0 && wave >= 4) __builtin_amdgcn_s_sleep(MVSN_WN_STAGGER);
        WN_STAMP();   // barrier
        xf_prepare();      // step + 2: its scalar loads travel behind this step's multiplies
      }
      rd_issue(flat, chunk);
      // multiplies of `step` with the transform of `step + 1` slotted between them
      // (a half that runs as a burst fetches its own patch right in front of its multiplies and consumes it right
      // behind them: the two halves' patches share their 16 registers)
      float dn[KS][4][4];
      if (has_next && !(MVSN_WN_ABLATE & 32)) {
        if (MVSN_WN_XF == 1) tr_load(dn);
        else if (MVSN_WN_XF == 0) tr_load(dn, 0, 1);
      }
      auto multiply = [&](auto first, auto hc) {   // one k-step: 16 coefficient GEMMs x 2 cout tiles
        constexpr bool FIRST = decltype(first)::value;
        constexpr int h = decltype(hc)::value;
        const float *ub = U + (VOL ? mm_stage : chunk) * UST + h * WN_UFLOATS + lane;
        float fb[2][2];
        fb[0][0] = ub[0], fb[0][1] = ub[64];
        // The input transform of step + 1 (B^T d B of the patch tr_load fetched) is interleaved with the multiplies,
        // two VALU instructions per coefficient: a row of the column transform t one row ahead, and -- right after
        // the MFMAs that read v[h][xi] for the last time -- the coefficient of step + 1 that replaces it.  Issued
        // after the burst the 64 instructions of the SIMD's two waves ran with the matrix pipe idle.
        auto tcol = [&](int i, int j) {
          const float d0 = dn[h][0][dslot(j)], d1 = dn[h][1][dslot(j)], d2 = dn[h][2][dslot(j)], d3 = dn[h][3][dslot(j)];
          return i == 0 ? d0 - d2 : (i == 1 ? d1 + d2 : (i == 2 ? d2 - d1 : d1 - d3));
        };
        // INTER: which k-step halves carry that interleave; the others run their 32 MFMAs as one burst and transform
        // behind it (MVSN_WN_XF: 0 = first half interleaved, second as a burst; 1 = both interleaved; 2 = both bursts)
        constexpr bool INTER = MVSN_WN_XF == 1 || (MVSN_WN_XF == 0 && h == 0);
        float tc[4], tn[4];
        if constexpr (INTER) {
#pragma unroll
          for (int j = 0; j < 4; ++j) tc[j] = tcol(0, j);
        } else {
          if (has_next && !(MVSN_WN_ABLATE & 32)) tr_load(dn, h, h + 1);
        }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {   // next coefficient's U fragments in flight behind this one's MFMAs
          const int cur = xi & 1, ti = xi >> 2, tj = xi & 3;
          if (INTER && ti < 3) tn[tj] = tcol(ti + 1, tj);
          if (xi + 1 < 16 && !(MVSN_WN_ABLATE & 2)) {
            fb[cur ^ 1][0] = ub[(xi + 1) * 128];
            fb[cur ^ 1][1] = ub[(xi + 1) * 128 + 64];
          }
          // TR (dilation 1): operands swapped -- D = couts x patches, a lane holds 4 consecutive couts of ONE patch,
          // so the 16 lanes of a row store 128 contiguous bytes of an output row (see finish_tile_tr).  The fragments
          // are the same registers either way (A and B of 16x16x4 share their lane layout).
          if constexpr (FIRST) {
            acc[xi][0] = TR ? mfma16x16x4(fb[cur][0], v[h][xi], floatx4{0.f, 0.f, 0.f, 0.f})
                            : mfma16x16x4(v[h][xi], fb[cur][0], floatx4{0.f, 0.f, 0.f, 0.f});
            acc[xi][1] = TR ? mfma16x16x4(fb[cur][1], v[h][xi], floatx4{0.f, 0.f, 0.f, 0.f})
                            : mfma16x16x4(v[h][xi], fb[cur][1], floatx4{0.f, 0.f, 0.f, 0.f});
          } else {
            acc[xi][0] = TR ? mfma16x16x4(fb[cur][0], v[h][xi], acc[xi][0]) : mfma16x16x4(v[h][xi], fb[cur][0], acc[xi][0]);
            acc[xi][1] = TR ? mfma16x16x4(fb[cur][1], v[h][xi], acc[xi][1]) : mfma16x16x4(v[h][xi], fb[cur][1], acc[xi][1]);
          }
          if (h == 0 && xi == 1 && has_next && !(MVSN_WN_ABLATE & 4)) {   // behind the first MFMAs:
            pf_issue();   // raw tile of step + NSTAGE into the stage `step` released
            uq_issue();   // VOL: U of step + NSTAGE - 1 into the slot step - 1 released
          }
          if constexpr (INTER) {
            if (!(MVSN_WN_ABLATE & 1)) {   // (without a next step dn is undefined and v is never read again)
              v[h][xi] = tj == 0 ? tc[0] - tc[2] : (tj == 1 ? tc[1] + tc[2] : (tj == 2 ? tc[2] - tc[1] : tc[1] - tc[3]));
              // pinned here: the coefficient is only USED a step later, and the compiler's sinking pass otherwise moves
              // the arithmetic towards that use -- out of this half, across the scheduling barriers (seen twice in round 4)
              if (MVSN_WN_PIN) asm volatile("" : "+v"(v[h][xi]));
            }
            if (tj == 3) {
#pragma unroll
              for (int k = 0; k < 4; ++k) tc[k] = tn[k];
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // keep the interleaving as written (and the live ranges short)
        }
        if constexpr (!INTER) {
          if (!(MVSN_WN_ABLATE & 1)) {
            tr_finish(dn, v, h, h + 1);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
              if (MVSN_WN_PIN) asm volatile("" : "+v"(v[h][xi]));   // (pinned, as above)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      multiply(firstc, std::integral_constant<int, 0>{});
      // Odd chunk counts (the 36-channel heads): the second half of the last step multiplies zero tiles (channels past
      // cin are fetched from the zero line) with the zero chunk the prologue put behind U.  As an if / else -- "nothing in
      // the second half: only finish the transform" -- the compiler hoisted / sank the coefficient arithmetic the two arms
      // had in common out of EVERY instantiation's second half: its 32 VALU instructions (+ 16 register copies) ran
      // behind the MFMA burst, on both waves of a SIMD at once, with the matrix pipe idle -- ~1 k of the 6 k cycles of
      // a step (s_memtime stamps, round 4).
      if constexpr (KS == 2) multiply(std::false_type{}, std::integral_constant<int, 1>{});
      tr_advance();
      if constexpr (VOL) mm_stage = mm_stage + 1 == NSTAGE ? 0 : mm_stage + 1;
      WN_STAMP();   // MFMAs issued + next transform
      ++step;
    };
    do_step(std::true_type{});
    for (chunk = 1; chunk < nsteps; ++chunk) do_step(std::false_type{});
    if (!(MVSN_WN_ABLATE & 16) || n < 0) {
      if constexpr (TR) finish_tile_tr(n, z, tile_id, y0, x0, r_s);
      else finish_tile(n, z, tile_id, y0, x0);
    }
  }
  if constexpr (RIDE > 0) {
    if (total_steps > 0) {   // the last step's units
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int ln;   // (re-derived: `lane` kept alive across the tile loop for this one use was spilled around it)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      rd_consume(ln);
    }
  }
#ifdef MVSN_WN_STAMPS
  if (dbg)
    for (int i = 0; i < dbg_i; ++i) dbg[i] = dbg_lds[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// 5 x 5 stride-2 32 -> 32 layers (the extractor's downsampling convolutions, multi_view_stereonet.py:78-129) as
// Winograd F(2x2,3x3) on the four stride-2 phases of the input: with P_{py,px}(r, c) = in(2r + py, 2c + px),
//   out(oy, ox) = sum_{py,px} sum_{a,b} g_{py,px}[a][b] * P_{py,px}(oy + a - 1, ox + b - 1),
//   g_{py,px}[a][b] = w[2a + py][2b + px]   (zero where 2a + py = 5 or 2b + px = 5),
// i.e. a 3 x 3 stride-1 pad-1 layer with 128 input channels.  Per output and (cin, cout) pair 16 products per 2 x 2
// patch and phase; an odd phase's third tap row / column is zero, so are its coefficients xi = (3, .) / (., 3): 49 of
// the 64 coefficient GEMMs remain -- 392 multiplies per 2 x 2 outputs and channel pair instead of the direct form's
// 800 (x 0.49).  The raw tile stays in the image's own layout (fetched by descriptor LDS-DMA like conv_wino_kernel's); a
// phase is the same tile sampled at stride 2 from (py, px) -- what a dilation-2 layer's patches do.
//
// Persistent workgroups of 8 waves, one per CU; tile = 16 x 32 outputs (8 x 16 patches, wave = patch row); a step =
// 4 input channels x 4 phases (98 MFMAs per wave) on one raw stage (4 channels x 35 rows x 72 columns) and that
// step's 32 KB of U, both double-buffered, one barrier per step.
constexpr int S2_TY = 16, S2_TX = 32;                   // output tile
constexpr int S2_ROWS = 2 * S2_TY + 3, S2_XS = 2 * S2_TX + 8, S2_DQ = S2_XS / 4;   // raw tile: rows 2 y0 - 2 .., columns 2 x0 - 4 ..
constexpr int S2_GROUPS = S2_ROWS * S2_DQ, S2_PIECES = (S2_GROUPS + 63) / 64;       // 630 groups, 10 pieces per channel
constexpr int S2_RCST = S2_ROWS * S2_XS + 1;            // channel stride = 1 (mod 4): the four channels of a k-step on disjoint banks
constexpr int S2_STAGE = 4 * S2_RCST;                   // raw stage (floats)
constexpr int S2_UST = 4 * WN_UFLOATS;                  // U of a step: [phase][xi][cout tile][lane]
constexpr int S2_STEPS = 8;                             // steps per tile (32 input channels)
constexpr size_t S2_LDS_BYTES = (size_t)(2 * S2_STAGE + 2 * S2_UST + 32) * sizeof(float);
static_assert(S2_PIECES * 4 % WN_WAVES == 0, "a step's raw pieces divide evenly among the waves");
static_assert(S2_LDS_BYTES <= 160 * 1024, "LDS plan");

__global__ void wino_s2_pack_kernel(const float *__restrict__ w, float *__restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S2_STEPS * S2_UST) return;
  const int lane = idx & 63, t = (idx >> 6) & 1, xi = (idx >> 7) & 15, ph = (idx >> 11) & 3, cc = idx >> 13;
  const int py = ph >> 1, px = ph & 1;
  const int co = t * 16 + (lane & 15), ci = cc * 4 + (lane >> 4);
  const float *g = w + ((size_t)co * 32 + ci) * 25;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const int i = xi >> 2, j = xi & 3;
  float u = 0.0f;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const int ky = 2 * a + py, kx = 2 * b + px;
      if (ky < 5 && kx < 5) u += G[i][a] * g[ky * 5 + kx] * G[j][b];
    }
  out[idx] = u;
}

#ifndef MVSN_S2_PIN
#define MVSN_S2_PIN 1
#endif
// MVSN_S2_CNTWAIT: the counted wait at the top of a tile's first step (see the step loop).  Kept on: 1-3 % faster than a
// full drain (profiles/r05_slab/s2_counted_wait_ab.txt); what it leans on -- at least eight vector-memory instructions
// between a step's last DMA piece and the wait on every path that ran a tile epilogue -- is asserted on the compiled
// code by tools/check_dma_isa.py (CPU test tests/test_dma_isa_cpu.py).
#ifndef MVSN_S2_CNTWAIT
#define MVSN_S2_CNTWAIT 1
#endif
struct WinoS2Args {
  int n, H, W, Ho, Wo, ntx, tiles;
  WinoDiv fd_ntx;
};

__global__ __launch_bounds__(WN_THREADS, 2) void conv_wino_s2_kernel(WinoS2Args g, const float *__restrict__ in,
                                                                     const float *__restrict__ upk,
                                                                     const float *__restrict__ bias,
                                                                     float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *U = smem + 2 * S2_STAGE;
  float *bias_lds = U + 2 * S2_UST;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t plane = (size_t)g.H * g.W;
  const int oplane = g.Ho * g.Wo;
  const int total = g.n * g.tiles, G = gridDim.x;
  const int slot = xcd_tile_index(blockIdx.x, G);
  const int my_items = slot < total ? (total - slot + G - 1) / G : 0;
  const int total_steps = my_items * S2_STEPS;
  if (tid < 32) bias_lds[tid] = bias ? bias[tid] : 0.0f;   // published by the first barrier

  // ---- fetch side: wave w takes pieces (w & 1) * 5 .. + 4 of channel w >> 1 of the step, and 4 KB of its U
  constexpr int PER = S2_PIECES * 4 / WN_WAVES;   // 5
  const int dch = wave >> 1, dp0 = (wave & 1) * PER;
  int pf_item = 0, pf_cc = 0, pf_n = 0;
  int pf_goff[PER];
  auto pf_plan = [&]() {
    const int flat = pf_item * G + slot;
    const int n = flat / g.tiles, tile = flat - n * g.tiles;
    const int tyi = wdiv(tile, g.fd_ntx), txi = tile - tyi * g.ntx;
    const int gy0 = 2 * tyi * S2_TY - 2, gx0 = 2 * txi * S2_TX - 4;
    pf_n = n;
    int lo;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = (dp0 + i) * 64 + lo;
      const int row = e / S2_DQ, q = e - row * S2_DQ;
      const int gy = gy0 + row, gx = gx0 + 4 * q;
      // rows outside the image fall outside the plane's descriptor too; columns would wrap into a neighbouring row
      pf_goff[i] = (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) ? (gy * g.W + gx) * 4 : -1;
    }
  };
  auto pf_issue = [&](int step) {   // DMA of this workgroup's step `step` into stage step & 1
    if (step >= total_steps) return;
    const float *src = in + ((size_t)pf_n * 32 + pf_cc * 4 + dch) * plane;
    float *dst = smem + (step & 1) * S2_STAGE + dch * S2_RCST;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if ((dp0 + i) * 64 + lane < S2_GROUPS)   // lanes past the tile's last group stay out of the slot
        wn_dma16_buf<true>(src, (unsigned)plane * 4u, (unsigned)pf_goff[i], dst + (dp0 + i) * 256);
    const float *us = upk + (size_t)pf_cc * S2_UST + wave * 1024 + lane * 4;
    float *ud = U + (step & 1) * S2_UST + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) wn_dma16<true>(us + i * 256, ud + i * 256);
    if (++pf_cc == S2_STEPS) {
      pf_cc = 0;
      ++pf_item;
      if (pf_item < my_items) pf_plan();
    }
  };
  if (my_items > 0) pf_plan();
  pf_issue(0);

  const int pcol = lane & 15, kc = lane >> 4;
  // this lane's patch (row wave, column pcol) of channel kc: phase (0, 0) sample (0, 0) inside a raw stage
  const int rbase = kc * S2_RCST + (4 * wave) * S2_XS + 4 * pcol + 2;

  floatx4 acc[16][2];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) acc[xi][0] = acc[xi][1] = floatx4{0.f, 0.f, 0.f, 0.f};

  int item = 0, cc = 0;
  for (int step = 0; step < total_steps; ++step) {
    // this wave's pieces of the step have landed.  (Behind them the wave has issued at most the eight stores of the tile
    // it just finished: vector memory instructions retire in issue order, see wait_landed of conv_wino_kernel.)
    if (MVSN_S2_CNTWAIT && cc == 0 && step > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wn_barrier<true>();                                  // ... everyone's; everyone is done with the other stage
    pf_issue(step + 1);
    const float *raw = smem + (step & 1) * S2_STAGE + rbase;
    const float *us = U + (step & 1) * S2_UST + lane;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      // the phases (py, 0) and (py, 1) together: their 4 x 4 samples at stride 2 are neighbouring columns of the raw
      // tile -- one ds_read2_b32 per pair, the input transform B^T d B on (px = 0, px = 1) register pairs (v_pk_add_f32).
      // The row an odd phase only multiplies by zero taps is not read; its last column is (its coefficients are skipped).
      typedef float s2_f2 __attribute__((ext_vector_type(2)));
      s2_f2 d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float *q = raw + (py + 2 * i) * S2_XS + 2 * j;
          d[i][j] = (py && i == 3) ? s2_f2{0.f, 0.f} : s2_f2{q[0], q[1]};
        }
      s2_f2 t[4][4], v[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i * 4 + 0] = t[i][0] - t[i][2];
        v[i * 4 + 1] = t[i][1] + t[i][2];
        v[i * 4 + 2] = t[i][2] - t[i][1];
        v[i * 4 + 3] = t[i][1] - t[i][3];
      }
      // (pinned as register pairs: with the odd column phase's unused halves visible the compiler narrows half of the
      // packed adds to scalar ones and shuffles the pairs with moves -- 200 VALU instructions per step instead of ~60)
      if (MVSN_S2_PIN) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
          if (!(py && (xi >> 2) == 3)) asm volatile("" : "+v"(v[xi]));
      }
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        if (py && (xi >> 2) == 3) continue;   // U_xi = 0 for the odd row phase
        const float *u = us + ((py * 2) * 16 + xi) * 128;
        acc[xi][0] = mfma16x16x4(v[xi].x, u[0], acc[xi][0]);
        acc[xi][1] = mfma16x16x4(v[xi].x, u[64], acc[xi][1]);
        if ((xi & 3) != 3) {                  // ... and for the odd column phase
          acc[xi][0] = mfma16x16x4(v[xi].y, u[16 * 128], acc[xi][0]);
          acc[xi][1] = mfma16x16x4(v[xi].y, u[16 * 128 + 64], acc[xi][1]);
        }
      }
    }
    if (++cc == S2_STEPS) {
      cc = 0;
      // ---- tile epilogue: Y = A^T m A per (patch, cout); lane: cout t * 16 + (lane & 15), patches 4 kc .. + 3 of
      // patch row `wave` = output columns 8 kc .. + 7 of rows 2 wave, 2 wave + 1: two 16-byte stores per row
      const int flat = item * G + slot;
      const int n = flat / g.tiles, tile = flat - n * g.tiles;
      const int tyi = wdiv(tile, g.fd_ntx), txi = tile - tyi * g.ntx;
      const int y0 = tyi * S2_TY + 2 * wave, x0 = txi * S2_TX;
      const __amdgpu_buffer_rsrc_t osrd = wn_rsrc(out + (size_t)n * 32 * oplane, (unsigned)(32 * oplane) * 4u);
      int lo;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
      const int ocl = lo & 15, oq = lo >> 4;
      unsigned ovoff[2];
#pragma unroll
      for (int hx = 0; hx < 2; ++hx)   // Wo % 4 == 0: four columns or none
        ovoff[hx] = x0 + 8 * oq + 4 * hx < g.Wo ? (unsigned)((ocl * oplane + 8 * oq + 4 * hx) * 4) : 0xFFFFFFFFu;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const float bv = bias_lds[tt * 16 + ocl];
        float y[4][2][2];   // [patch r][row a][column b]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float tm[2][4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            tm[0][j] = acc[0 + j][tt][r] + acc[4 + j][tt][r] + acc[8 + j][tt][r];
            tm[1][j] = acc[4 + j][tt][r] - acc[8 + j][tt][r] - acc[12 + j][tt][r];
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            y[r][a][0] = tm[a][0] + tm[a][1] + tm[a][2] + bv;
            y[r][a][1] = tm[a][1] - tm[a][2] - tm[a][3] + bv;
          }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {   // (a row past the image: the same stores with out-of-range offsets -- always eight per tile)
          const bool rok = y0 + a < g.Ho;   // uniform
          const unsigned so = rok ? (unsigned)((tt * 16 * oplane + (y0 + a) * g.Wo + x0) * 4) : 0u;
          wn_store4<0>(osrd, rok ? ovoff[0] : 0xFFFFFFFFu, so, floatx4{y[0][a][0], y[0][a][1], y[1][a][0], y[1][a][1]});
          wn_store4<0>(osrd, rok ? ovoff[1] : 0xFFFFFFFFu, so, floatx4{y[2][a][0], y[2][a][1], y[3][a][0], y[3][a][1]});
        }
      }
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) acc[xi][0] = acc[xi][1] = floatx4{0.f, 0.f, 0.f, 0.f};
      ++item;
    }
  }
}

bool wino_geom(const mvsn_conv_desc *d, WinoGeom *g) {
  if (!d || d->precision != MVSN_CONV_FP32_WINO) return false;
  if (d->n <= 0 || d->c_in <= 0 || d->c_out != 32 || d->depth < 1 || d->rows <= 0 || d->cols <= 0) return false;
  g->s2 = false;
  if (d->kd == 1 && d->kh == 5 && d->kw == 5 && d->stride == 2) {   // 5 x 5 stride 2 on the input's four phases (conv_wino_s2_kernel)
    if (d->c_in != 32 || d->dilation != 1 || d->depth != 1 || d->cols % 8 != 0) return false;
    if ((unsigned long long)d->rows * d->cols * 32ull * 4ull >= (1ull << 31)) return false;   // 32-bit descriptor offsets
    g->s2 = true, g->wide = 0;
    g->n = d->n, g->cin = 32, g->H = d->rows, g->W = d->cols, g->dil = 1, g->D = 1, g->vol = false;
    g->nty = ((d->rows - 1) / 2 + 1 + S2_TY - 1) / S2_TY;
    g->ntx = ((d->cols - 1) / 2 + 1 + S2_TX - 1) / S2_TX;
    g->tiles = g->nty * g->ntx;
    g->nchunks = S2_STEPS * 4;
    g->packed_floats = (size_t)S2_STEPS * S2_UST;
    return (long)g->n * g->tiles < (1L << 30);
  }
  if ((d->kd != 1 && d->kd != 3) || d->kh != 3 || d->kw != 3 || d->stride != 1) return false;
  if (d->kd == 1 && d->depth != 1) return false;
  if (d->dilation != 1 && d->dilation != 2 && d->dilation != 4 && d->dilation != 8) return false;
  if (d->cols % 4 != 0) return false;
  // tiles travel through buffer descriptors with 32-bit byte offsets: a sample's 32 channels must stay below 4 GB
  if ((unsigned long long)d->depth * d->rows * d->cols * 32ull * 4ull >= (1ull << 32)) return false;
  g->n = d->n, g->cin = d->c_in, g->H = d->rows, g->W = d->cols, g->dil = d->dilation;
  g->D = d->depth, g->vol = d->kd == 3;
  g->nty = (d->rows + WN_TY - 1) / WN_TY;
  g->ntx = (d->cols + WN_TX - 1) / WN_TX;
  // volume form on planes a little wider than one 16 x 32 tile: 10-row strips of the whole width (WIDE) wherever that
  // takes fewer tile slots per plane (30 x 40: 3 strips of 100 patches instead of 2 x 2 tiles of 128 slots)
  g->wide = 0;
#ifndef MVSN_WN_NO_WIDE
  if (d->kd == 3 && d->cols > WN_TX && d->cols <= WN_WIDE_TX) {
    const int strips = (d->rows + WN_WIDE_TY - 1) / WN_WIDE_TY;
    if (strips < g->nty * g->ntx) g->wide = 1, g->nty = strips, g->ntx = 1;
  }
#endif
  g->tiles = g->nty * g->ntx;
#ifndef MVSN_WN_NO_ROLL
  // ... or, where that takes fewer items still, six patch rows at a time rolling through the sample's planes (WIDE 2):
  // `tiles` then counts the items of a SAMPLE (30 x 40 x D: 2.5 D instead of 3 D)
  if (d->kd == 3 && d->cols > WN_TX && d->cols <= WN_WIDE_TX && d->c_in == 32 && d->dilation == 1) {
    const int pr = (d->rows + 1) / 2;
    const long items = ((long)d->depth * pr + WN_ROLL_PR - 1) / WN_ROLL_PR;
    if (pr >= WN_ROLL_PR && items < (long)d->depth * g->tiles && items < (1L << 24))
      g->wide = 2, g->nty = 1, g->ntx = 1, g->tiles = (int)items;
  }
#endif
  if (g->vol) {   // volume form: 32 -> 32 channels, dilation 1; U streams, 3 x 8 chunks
    if (d->c_in != 32 || d->dilation != 1) return false;
    if ((long)wino_items(*g) > 1L << 24) return false;
    g->nchunks = 24;
    g->packed_floats = (size_t)g->nchunks * WN_UFLOATS;
    return true;
  }
  g->nchunks = (d->c_in + 3) / 4;
  if (g->nchunks > WN_MAX_CHUNKS) return false;   // U must stay resident in LDS
  if (g->nchunks > 8 && d->dilation > 2) return false;   // ... next to the (larger) raw-tile ring of dilation 4 / 8
  if (g->nchunks == 1 && d->dilation != 1) return false;  // dilated 4-channel layers are not instantiated
  g->packed_floats = (size_t)g->nchunks * WN_UFLOATS;
  return true;
}

// (cout, cin, 3, 3) -> the [chunk][xi][cout tile][lane] layout above, without a descriptor (the chain's stepwise form)
int wino_pack_2d(const float *weight, int cin, float *packed, hipStream_t stream) {
  const int nchunks = (cin + 3) / 4, total = nchunks * WN_UFLOATS;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, weight, cin, 32, nchunks, 1,
                     packed);
  return check_launch("wino_pack_2d");
}

int wino_pack(const mvsn_conv_desc *d, const float *weight, float *packed, hipStream_t stream) {
  WinoGeom g;
  if (!wino_geom(d, &g)) return MVSN_E_BADARG;
  if (g.s2) {
    hipLaunchKernelGGL(wino_s2_pack_kernel, dim3((S2_STEPS * S2_UST + 255) / 256), dim3(256), 0, stream, weight, packed);
    return check_launch("mvsn_conv_pack_weights(winograd, 5x5 stride 2)");
  }
  const int total = g.nchunks * WN_UFLOATS;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, weight, g.cin, 32, g.nchunks,
                     g.vol ? 3 : 1, packed);
  return check_launch("mvsn_conv_pack_weights(winograd)");
}

// RIDE units per wave and step of the instantiation a layer runs on (0: that kernel carries nothing)
static int wino_ride_units(const WinoGeom &g) {
  if (g.s2) return 0;
  if (g.vol) return 1;         // 12 steps per (plane, tile): 96 units, of which a job of the layer's own size needs 64
  if (g.nchunks != 8) return 0;
  // 4 steps of two k-steps / 8 steps of one: 64 units = 64 KB per tile either way.  Dilation 4 carries on the
  // one-k-step form (its two-k-step ring, 94 KB next to 64 KB of U, leaves no room for the residual slots).
  return g.dil >= 4 ? 1 : 2;
}

bool wino_can_carry(const WinoGeom &g, const mvsn_apply_job *job) {
  const int r = wino_ride_units(g);
  if (!r || !job || !job->x || !job->stats || !job->gamma || !job->beta || !job->out || job->n <= 0) return false;
  if (job->spatial <= 0 || job->spatial % (256 * r) != 0) return false;   // a step's units lie in one plane
  if (job->r_stats && !(job->residual && job->r_gamma && job->r_beta)) return false;
  if ((((size_t)job->x | (size_t)job->out | (size_t)job->residual) & 15) != 0) return false;
  const long units = (long)job->n * 32 * (job->spatial / 256);
  if (g.vol && job->residual) return false;   // (no residual slots next to the volume form's two rings)
  const int nsteps = g.vol ? 12 : (g.dil >= 4 ? 8 : 4);
  const long capacity = (long)g.n * wino_items(g) * nsteps * WN_WAVES * r;   // unit indices are 32-bit in the kernel
  return units <= capacity && capacity < (1L << 31) && units < (1L << 30);
}

int wino_launch(const WinoGeom &g, const float *in, const float *upk, const float *bias, const float *in_stats,
                const float *in_gamma, const float *in_beta, float *out, float *out_partials, hipStream_t stream,
                const WinoBlocks *blocks, const mvsn_apply_job *job) {
  if (g.s2) {
    if (blocks || job || in_stats || out_partials) {
      set_error("mvsn_conv_forward(winograd, 5x5 stride 2): no input transform, statistics, channel blocks or carried job");
      return MVSN_E_BADARG;
    }
    WinoS2Args a;
    a.n = g.n, a.H = g.H, a.W = g.W, a.Ho = (g.H - 1) / 2 + 1, a.Wo = (g.W - 1) / 2 + 1, a.ntx = g.ntx, a.tiles = g.tiles;
    a.fd_ntx = wino_div((unsigned)g.ntx);
    static LdsOptIn opt;
    if (int rc = ensure_lds(opt, (const void *)conv_wino_s2_kernel, S2_LDS_BYTES, "mvsn_conv_forward(winograd, 5x5 stride 2)"))
      return rc;
    const long total = (long)g.n * g.tiles;
    const int cus = device_cus();
    hipLaunchKernelGGL(conv_wino_s2_kernel, dim3((unsigned)(total < cus ? total : cus)), dim3(WN_THREADS), S2_LDS_BYTES,
                       stream, a, in, upk, bias, out);
    return check_launch("mvsn_conv_forward(winograd, 5x5 stride 2)");
  }
  RideArgs rd = {};
  if (job) {
    if (blocks || !wino_can_carry(g, job)) {
      set_error("mvsn_conv_forward(winograd): this launch cannot carry the job");
      return MVSN_E_BADARG;
    }
    rd.x = job->x, rd.stats = job->stats, rd.gamma = job->gamma, rd.beta = job->beta;
    rd.res = job->residual, rd.r_stats = job->r_stats, rd.r_gamma = job->r_gamma, rd.r_beta = job->r_beta;
    rd.out = job->out;
    rd.units = (int)((long)job->n * 32 * (job->spatial / 256));
    rd.fd_upp = wino_div((unsigned)(job->spatial / 256));
  }
  WinoArgs a;
  a.n = g.n, a.cin = g.cin, a.H = g.H, a.W = g.W, a.ntx = g.ntx, a.tiles = g.tiles, a.nchunks = g.nchunks;
  a.fd_ntx = wino_div((unsigned)g.ntx);
  a.cb0 = g.cin, a.cb1 = 0, a.in1 = a.in2 = in;
  a.D = g.D;
  a.pr = (g.H + 1) / 2, a.fd_pr = wino_div((unsigned)a.pr);
  if (blocks) a.cb0 = blocks->cb0, a.cb1 = blocks->cb1, a.in1 = blocks->in1, a.in2 = blocks->in2;
  a.rev = (job && job->reverse == 1) ? 1 : 0;
  a.cs0 = (size_t)g.H * g.W, a.bs0 = (size_t)a.cb0 * a.cs0;
  if (blocks && blocks->cs0) a.cs0 = blocks->cs0, a.bs0 = blocks->bs0;
  const int cus = device_cus();
  // (k-steps per step, ring depth) by what fits next to the resident U: the raw tile grows with the dilation
  //   dilation 1: 2 k-steps x 4 stages (94 KB); 2, 4: 2 x 3 (78 / 94 KB); 8: 1 x 3 (74 KB); 4-channel head: 1 x 6
  //   volume form: 2 x 3 raw stages + 3 stages of U (118 KB)
  const bool head = g.nchunks == 1;
  const int ks = (head || g.dil == 8 || (job && g.dil == 4)) ? 1 : 2;
  // (a carrying launch waits with one step of DMA in flight, see RideArgs: a fourth stage would never be used)
  // ... except the rolling strips without a job: four stages (158 KB) measured 3.84-3.87 against 3.90-3.92 ms per 128
  // samples of 96 x 30 x 40 (the 16 x 32 tiles: 0.992 either way, they stay at three)
  const int nstage = head ? 6 : (((g.dil == 1 && g.nchunks <= 8 && !job) || (g.vol && g.wide == 2 && !job)) ? 4 : 3);
  const size_t rcst = g.wide ? (size_t)(g.wide == 2 ? WN_ROLL_HY : WN_WIDE_TY + 2) * (WN_WIDE_TX + 2 * wn_pa(1)) + 16
                             : (size_t)wn_rcst(g.dil);
  size_t lds = ((size_t)nstage * ks * 4 * rcst +
                (g.vol ? (size_t)nstage * ks : (size_t)((g.nchunks + ks - 1) / ks * ks)) * WN_UFLOATS) * sizeof(float);
  lds += 32 * sizeof(float);                                      // bias
  if (job && !g.vol) lds += (size_t)WN_WAVES * wino_ride_units(g) * 1024;   // the carried job's residual slots
#ifdef MVSN_WN_STAMPS
  lds += 1024;   // stamp area
#endif
  if (head && g.dil != 1) {
    set_error("mvsn_conv_forward(winograd): dilated 4-channel layers are not instantiated");
    return MVSN_E_BADARG;
  }
  if (lds > 160 * 1024) {
    set_error("mvsn_conv_forward(winograd): %zu bytes of LDS needed", lds);
    return MVSN_E_TOOLARGE;
  }
  dim3 grid(1);
#define WN_CASE(M, K, N, D, ...)                                                                                   \
  do {                                                                                                             \
    static LdsOptIn opt;                                                                                           \
    if (int rc = ensure_lds(opt, (const void *)conv_wino_kernel<M, K, N, D, ##__VA_ARGS__>, lds,                   \
                            "mvsn_conv_forward(winograd)"))                                                        \
      return rc;                                                                                                   \
    hipLaunchKernelGGL((conv_wino_kernel<M, K, N, D, ##__VA_ARGS__>), grid, dim3(WN_THREADS), lds, stream, a, in,  \
                       upk, bias,                                                                                  \
                       in_stats, in_gamma, in_beta, out, out_partials, rd, (const void *)a.in1, (const void *)a.in2, \
                       (const void *)rd.x, (const void *)rd.stats, (const void *)rd.res, (const void *)rd.r_stats,  \
                       (const void *)rd.out, (const void *)nullptr, (const void *)nullptr, (const void *)nullptr);  \
  } while (0)
  const long total = (long)g.n * wino_items(g);
  grid = dim3((unsigned)(total < cus ? total : cus));   // persistent: one workgroup per CU walks items grid-strided
  const bool xf = in_stats != nullptr;
  // volume form on planes one tile wide (the 16 x 32 coarse grid): neighbouring lanes share the input transform's column
  // sums (FULLW, see conv_wino_kernel)
  const bool fullw = MVSN_WN_FULLW_OK && g.vol && !g.wide && g.ntx == 1 && g.W <= WN_TX;
  if (job) {   // the same kernels with the carried job's loads / stores in their steps
    if (g.vol && g.wide == 2) { if (xf) WN_CASE(1, 2, 3, 1, true, 1, 2); else WN_CASE(0, 2, 3, 1, true, 1, 2); }
    else if (g.vol && g.wide) { if (xf) WN_CASE(1, 2, 3, 1, true, 1, 1); else WN_CASE(0, 2, 3, 1, true, 1, 1); }
    else if (g.vol && fullw) { if (xf) WN_CASE(1, 2, 3, 1, true, 1, 0, true); else WN_CASE(0, 2, 3, 1, true, 1, 0, true); }
    else if (g.vol) { if (xf) WN_CASE(1, 2, 3, 1, true, 1); else WN_CASE(0, 2, 3, 1, true, 1); }
    else if (g.dil == 1) { if (xf) WN_CASE(1, 2, 3, 1, false, 2); else WN_CASE(0, 2, 3, 1, false, 2); }
    else if (g.dil == 2) { if (xf) WN_CASE(1, 2, 3, 2, false, 2); else WN_CASE(0, 2, 3, 2, false, 2); }
    else if (g.dil == 4) { if (xf) WN_CASE(1, 1, 3, 4, false, 1); else WN_CASE(0, 1, 3, 4, false, 1); }
    else { if (xf) WN_CASE(1, 1, 3, 8, false, 1); else WN_CASE(0, 1, 3, 8, false, 1); }
  } else
  if (g.vol && g.wide == 2) { if (xf) WN_CASE(1, 2, 4, 1, true, 0, 2); else WN_CASE(0, 2, 4, 1, true, 0, 2); }
  else if (g.vol && g.wide) { if (xf) WN_CASE(1, 2, 3, 1, true, 0, 1); else WN_CASE(0, 2, 3, 1, true, 0, 1); }
  else if (g.vol && fullw) { if (xf) WN_CASE(1, 2, 3, 1, true, 0, 0, true); else WN_CASE(0, 2, 3, 1, true, 0, 0, true); }
  else if (g.vol) { if (xf) WN_CASE(1, 2, 3, 1, true); else WN_CASE(0, 2, 3, 1, true); }
  else if (head) { if (xf) WN_CASE(1, 1, 6, 1); else WN_CASE(0, 1, 6, 1); }
  else if (g.dil == 1 && g.nchunks > 8) { if (xf) WN_CASE(1, 2, 3, 1); else WN_CASE(0, 2, 3, 1); }
  else if (g.dil == 1) { if (xf) WN_CASE(1, 2, 4, 1); else WN_CASE(0, 2, 4, 1); }
  else if (g.dil == 2) { if (xf) WN_CASE(1, 2, 3, 2); else WN_CASE(0, 2, 3, 2); }
  else if (g.dil == 4) { if (xf) WN_CASE(1, 2, 3, 4); else WN_CASE(0, 2, 3, 4); }
  else { if (xf) WN_CASE(1, 1, 3, 8); else WN_CASE(0, 1, 3, 8); }
#undef WN_CASE
  return check_launch("mvsn_conv_forward(winograd)");
}

}  // namespace mvsn

#ifdef MVSN_WN_STAMPS
extern "C" int mvsn_debug_set_wino_stamps(void *buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mvsn::g_wn_stamps), &buf, sizeof(buf));
}
#endif
