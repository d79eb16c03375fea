// K9: 3-D convolutions of VAE3DLarge (reference networks.py:2225-2264, 471-504) as implicit GEMM
// on MFMA.  Activations are NDHWC fp16, so the Cin channels of one input voxel are contiguous
// and an im2col row is a list of (tap, voxel) segments: the A-tile loader below gathers those
// segments with the same 16-B global_load_lds + source-side XOR swizzle as the dense GEMM
// (csrc/gemm_f16.hip); out-of-range taps (padding, K padding) read from a zero page.
// Conv3d (any k/stride/pad) and each output-parity class of ConvTranspose3d(k4,s2,p1) are the
// same kernel with different tap tables and output row maps.  Eval-mode BatchNorm3d is folded
// into W/bias on the host; epilogue = bias (+ residual) (+ ReLU).
//
// Also here: the two degenerate layers that are not GEMM shaped -- the first conv (Cin = 1,
// K = 27) and the last conv (Cout = 1) with its sigmoid -- as direct VALU kernels.
#include "common.h"

namespace pcd {

constexpr int CBK = 64;
constexpr int CROWB = CBK * 2;

constexpr int MAX_VARIANTS = 8;      // the 2x2x2 output-parity classes of a stride-2 ConvTranspose3d
constexpr int MAX_TAPS = 64;
constexpr int TAP_SLOTS = 128;     // LDS tap table of the implicit GEMM: ntaps real entries + the K padding's (kpad / Cin <= 128)

// what differs between the problems of one launch (blockIdx.y): tap table, weights, output parity
struct ConvVariant {
    const int* taps;                  // device [ntaps] packed (dz & 0xff) | (dy & 0xff) << 8 | (dx & 0xff) << 16
    const half_t* w;                  // [Cout][kpad]
    int pz, py, px, pad_;
};

struct ConvParams {
    const half_t* in; int D, H, W, Cin, cin_shift;
    int Do, Ho, Wo, stride;           // row space: m = ((b*Do+oz)*Ho+oy)*Wo+ox ; input coord = o*stride + d
    int ntaps, kpad;
    const float* bias;
    const half_t* resid;              // optional, indexed like out
    half_t* out; int Cout;
    int OD, OH, OW, os;               // output voxel = (oz*os+pz, oy*os+py, ox*os+px) in an OD x OH x OW grid
    int relu;
    int M;
    const half_t* zero;               // >= 128 B of zeros
    int tiles_n;
    int splits;                       // split-K (blockIdx.z); > 1 -> fp32 partials into slabs, conv3d_finish_kernel
    float* slabs;                     // [splits][nvar][M][Cout]
    int nvar;
    const half_t* in2;                // optional second source on the row grid (pcd_conv3d_desc_t.in2): K tiles kt >= kt2 read it
    int cin2_shift, kt2;              // log2(cin2); first K tile of the second source (INT_MAX without one)
    ConvVariant var[MAX_VARIANTS];
};

__device__ __forceinline__ void cglds16(const half_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same LDS-DMA from inline asm.  The compiler models the builtin as a FLAT access that may touch both
// memory and LDS, and while one is pending it turns every later s_waitcnt into vmcnt(0) / lgkmcnt(0) -- which
// drains the fragment prefetch of the halo kernel below.  Issued from asm, the DMA is invisible to that
// bookkeeping; the caller owns the vmcnt wait and the barrier.
__device__ __forceinline__ void cglds16_asm(const half_t* g, char* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void conv_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Before a barrier behind which ANOTHER wave's LDS-DMA refills the ring stage this wave has just read: the wave's own ds_reads must have
// RETURNED, not just been issued.  The compiler sinks a stage's last MFMAs (and the lgkmcnt wait in front of them) below the raw s_barrier,
// so a wave would pass it with fragment reads of that stage still queued in the LDS pipe; nothing orders them against the DMA's write, and with
// two workgroups per CU queueing reads and L1-resident weights coming back fast the write did win now and then (conv3d_k4s2_halo_kernel, batch 16:
// one encode in seven differed from the others by 1e-3..5e-3; tools/diag_vae_batch.py).  tools/check_barrier_reads.py scans the built code for it.
__device__ __forceinline__ void conv_reads_landed() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// K tile BKT (64 or 32 halfs: 128- or 64-byte LDS rows) in a ring of NST stages.  A stage is requested NST - 1 K tiles before
// the barrier that publishes it and NST - 2 younger stages stay in flight behind that barrier's counted vmcnt: with two
// stages the gather of K tile kt+1 has one K tile of MFMAs (~500 cycles per wave) to come back from L2, and the waves
// spend more than half their time at the wait (SQ_WAIT_ANY 56 % of SQ_WAVE_CYCLES on the 128 -> 128 layers).
// ABL: timing ablations (pcd_conv3d_config + 1024 x bits; OUTPUTS WRONG): 1 = the weight rows are staged for the first NST - 1 K tiles only, 2 = the same for the gathered rows,
// 4 = every fragment read takes the first fragment's address, 8 = no MFMAs
template <int BM, int BN, int BKT, int NST, int ABL = 0>
__global__ __launch_bounds__(256) void conv3d_igemm_kernel(ConvParams p) {
    constexpr int CBK = BKT, CROWB = BKT * 2;                 // shadow the file-level K tile
    constexpr int LPR = BKT / 8, RPW = 64 / LPR, RPI = 4 * RPW;   // lanes per staged row, rows per wave instruction / per round
    constexpr int KS = BKT / 32;
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    constexpr int STAGE_BYTES = (BM + BN) * CROWB;
    constexpr int OUT_LD = BN * 2 + 16;
    constexpr int LDS_BYTES = (NST * STAGE_BYTES > BM * OUT_LD) ? NST * STAGE_BYTES : BM * OUT_LD;
    constexpr int AR = BM / RPI, BR = BN / RPI, LPT = AR + BR;  // LPT: LDS-DMA instructions per thread and stage
    static_assert(ABL == 0 || NST == 2, "the ablations drop LDS-DMA instructions: only the two-stage form, whose waits are vmcnt(0)");
    static_assert(BM % RPI == 0 && BN % RPI == 0 && NST >= 2 && NST <= 4, "stage layout");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    __shared__ int4 taps_s[TAP_SLOTS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // consecutive workgroups go round-robin over the 8 XCDs: give each XCD a contiguous run of row tiles, so that the
    // tiles that gather the same input voxels (neighbouring output rows, all 27 taps) meet in one L2
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = p.kpad / BKT;
    const int cls = blockIdx.y, sp = blockIdx.z;
    const int kt0 = (int)((int64_t)sp * nk_all / p.splits), kt1 = (int)((int64_t)(sp + 1) * nk_all / p.splits);
    const ConvVariant& cv = p.var[cls];
    const half_t* __restrict__ wgt = cv.w;
    // Tap table -> LDS once, in the form the gather wants (the address arithmetic of a K tile is what bounds this
    // kernel's issue slots, not the MFMAs): x, y = the tap's (dz, dy, dx) as three biased byte fields, +64 + d and
    // 64 - d; z = its offset in elements of the NDHWC input.  A row keeps (z, y, x) + 64 and (D-1-z, H-1-y, W-1-x) + 64
    // in the same byte fields, so "input voxel inside the grid" is: bit 7 of every field of both sums set -- two
    // adds, two ands and one compare for all six bounds (dims <= 64, |d| < 64: no carry between fields).  Slots past
    // ntaps (K padding) hold 0 and fail that test.
    if (tid < TAP_SLOTS) {
        int4 e = {0, 0, 0, 0};
        if (tid < p.ntaps) {
            const int pk = cv.taps[tid];
            const int dz = (int)(signed char)(pk & 0xff), dy = (int)(signed char)((pk >> 8) & 0xff),
                      dx = (int)(signed char)((pk >> 16) & 0xff);
            e.x = (dz + 64) | (dy + 64) << 8 | (dx + 64) << 16;
            e.y = (64 - dz) | (64 - dy) << 8 | (64 - dx) << 16;
            e.z = ((dz * p.H + dy) * p.W + dx) * p.Cin;
        }
        taps_s[tid] = e;
    }
    __syncthreads();

    // the rows this thread stages (fixed for the whole K loop): decode the output voxel once
    const int srow = wave * RPW + lane / LPR;                // + r*RPI
    // logical 16-B chunk (same for every r): the XOR swizzles of gemm_f16.hip for 128- and 64-byte rows
    const int lchunk = (lane % LPR) ^ (BKT == 64 ? (srow >> 1) & 7 : (-(srow >> 2)) & 3);
    int rp1[AR], rp2[AR], rbase[AR];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        int m = m0 + r * RPI + srow;
        m = m < p.M ? m : p.M - 1;
        const int ox = m % p.Wo; int t = m / p.Wo;
        const int oy = t % p.Ho; t /= p.Ho;
        const int oz = t % p.Do;
        const int b = t / p.Do;
        const int rz = oz * p.stride, ry = oy * p.stride, rx = ox * p.stride;
        rp1[r] = (rz + 64) | (ry + 64) << 8 | (rx + 64) << 16;
        rp2[r] = (p.D - 1 - rz + 64) | (p.H - 1 - ry + 64) << 8 | (p.W - 1 - rx + 64) << 16;
        rbase[r] = (((b * p.D + rz) * p.H + ry) * p.W + rx) * p.Cin;
    }
    const half_t* wrow[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        int n = n0 + r * RPI + srow;
        n = n < p.Cout ? n : p.Cout - 1;
        wrow[r] = wgt + (int64_t)n * p.kpad;
    }
    const half_t* zlane = p.zero + (lane % LPR) * 8;

    // LDS-DMA from asm (the compiler would serialise every later wait behind the builtin form); the K loop owns vmcnt
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES;
        const int kidx = kt * CBK + lchunk * 8;
        const bool first = kt < kt0 + NST - 1;
        if ((ABL & 2) && !first) {
        } else
        if (kt >= p.kt2) {
            // second source (uniform per K tile): row m of in2, channels kidx - kt2 * CBK .. ; the row grid IS in2's grid (stride 1), so its
            // voxel index is rbase / Cin and no bound can fail
            const int c2 = kidx - p.kt2 * CBK;
#pragma unroll
            for (int r = 0; r < AR; ++r)
                cglds16_asm(p.in2 + (((rbase[r] >> p.cin_shift) << p.cin2_shift) + c2), base + (r * RPI + wave * RPW) * CROWB);
        } else {
            const int4 te = taps_s[kidx >> p.cin_shift];
            const int dc = te.z + (kidx & (p.Cin - 1));
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                const bool ok = (((rp1[r] + te.x) & (rp2[r] + te.y)) & 0x808080) == 0x808080;
                const half_t* g = p.in + (rbase[r] + dc);
                g = ok ? g : zlane;
                cglds16_asm(g, base + (r * RPI + wave * RPW) * CROWB);
            }
        }
        if ((ABL & 1) && !first) return;
#pragma unroll
        for (int r = 0; r < BR; ++r) cglds16_asm(wrow[r] + kidx, base + BM * CROWB + (r * RPI + wave * RPW) * CROWB);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ra = wm * WM + (lane & 15), rbw = wn * WN + (lane & 15);
    const int swa = BKT == 64 ? (ra >> 1) & 7 : (-(ra >> 2)) & 3, swb = BKT == 64 ? (rbw >> 1) & 7 : (-(rbw >> 2)) & 3;
    const int q = lane >> 4;
    int offa[KS], offb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        offa[ks] = ra * CROWB + (((ks * 4 + q) ^ swa) << 4);
        offb[ks] = BM * CROWB + rbw * CROWB + (((ks * 4 + q) ^ swb) << 4);
    }

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (kt0 + t < kt1) stage(kt0 + t, t);
    int buf = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        // stage kt must have landed; up to NST - 2 younger stages stay in flight
        const int younger = min(NST - 2, kt1 - 1 - kt);
        if (NST >= 4 && younger == 2) conv_wait_vmcnt<2 * LPT>();
        else if (NST >= 3 && younger == 1) conv_wait_vmcnt<LPT>();
        else conv_wait_vmcnt<0>();
        conv_reads_landed();
        __builtin_amdgcn_s_barrier();      // every wave's share of K tile kt is in LDS; the buffer of K tile kt-1 is free
        if (kt + NST - 1 < kt1) {
            int nb = buf + NST - 1;
            nb = nb >= NST ? nb - NST : nb;
            stage(kt + NST - 1, nb);
        }
        const char* base = smem + buf * STAGE_BYTES;
        buf = buf + 1 == NST ? 0 : buf + 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const half8*)(base + offa[ks] + ((ABL & 4) ? 0 : i * 16 * CROWB));
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *(const half8*)(base + offb[ks] + ((ABL & 4) ? 0 : j * 16 * CROWB));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr ((ABL & 8) != 0) acc[i][j][0] += (float)af[i][0] * (float)bf[j][0];
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue: bias -> LDS (fp16) -> row-contiguous 16-B stores (+ residual, ReLU) through the row map
    const int colq = lane & 15;
    if (p.splits > 1) {
        // split-K partial: raw fp32 accumulators to this split's slab; bias/residual/ReLU/row map in the finish
        float* slab = p.slabs + ((int64_t)sp * p.nvar + cls) * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * WN + j * 16 + colq;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WM + i * 16 + q * 4 + r;
                    if (m < p.M && col < p.Cout) slab[(int64_t)m * p.Cout + col] = acc[i][j][r];
                }
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int lcol = wn * WN + j * 16 + colq;
        const float bc = (p.bias != nullptr && n0 + lcol < p.Cout) ? p.bias[n0 + lcol] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = wm * WM + i * 16 + q * 4 + r;
                float v = acc[i][j][r] + bc;
                if (p.relu && p.resid == nullptr) v = fmaxf(v, 0.f);
                *(half_t*)(smem + lrow * OUT_LD + lcol * 2) = to_half_sat(v);
            }
    }
    __syncthreads();
    constexpr int CPR = BN / 8, TOTAL = BM * CPR;
#pragma unroll
    for (int it = 0; it < TOTAL / 256; ++it) {
        const int idx = it * 256 + tid;
        const int lrow = idx / CPR, ch = idx - lrow * CPR;
        const int m = m0 + lrow, col = n0 + ch * 8;
        if (m < p.M && col < p.Cout) {
            const int ox = m % p.Wo; int t = m / p.Wo;
            const int oy = t % p.Ho; t /= p.Ho;
            const int oz = t % p.Do; const int b = t / p.Do;
            const int64_t orow = (((int64_t)b * p.OD + oz * p.os + cv.pz) * p.OH + oy * p.os + cv.py) * p.OW +
                                 ox * p.os + cv.px;
            half8 v = *(const half8*)(smem + lrow * OUT_LD + ch * 16);
            if (p.resid != nullptr) {
                const half8 rs = *(const half8*)(p.resid + orow * p.Cout + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[e] + (float)rs[e];
                    if (p.relu) f = fmaxf(f, 0.f);
                    v[e] = to_half_sat(f);
                }
            }
            *(half8*)(p.out + orow * p.Cout + col) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// k3 / stride 1 / pad 1 with an LDS-resident input halo.  The implicit GEMM above re-gathers every input
// voxel once per tap (27x) from L2; at C_out <= 64 that gather traffic (24 KB per 1 MFLOP tile step) is
// what bounds the 32^3 layers.  Here a workgroup owns a 4 x 4 x 8 block of output voxels (128 GEMM rows),
// loads the 6 x 6 x 10 input halo ONCE into LDS and reads the A fragments of all 27 taps from it; only the
// weights stream (G taps per stage, LDS-DMA, three buffers).
// Halo image: voxel v = (hz * 6 + hy) * 10 + hx at v * C_in * 2 bytes, its 16-byte chunk c stored at chunk c ^ s(hx, hy):
//   C_in = 64 (8 chunks, two voxels per 256-B bank row):   s = ((hx >> 1) & 1) << 1 | (hy & 1) << 2
//   C_in = 32 (4 chunks, four voxels per bank row):        s = (hy & 1) << 1
// A ds_read_b128 is served in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS): a group holds all 16 rows of an MFMA block (two y rows of 8 x), rows 0-3 and 12-15 with
// k chunk q, rows 4-11 with q ^ 1, and is conflict-free when the 16 lanes hit 16 distinct 16-B slots of the bank row.
// With these s they do, for every tap shift (checked exhaustively; the round-1 layout, rows padded to C_in * 2 + 16
// bytes, was a 3-way conflict on every A read: SQ_LDS_BANK_CONFLICT 54 % of the LDS cycles).
struct HaloParams {
    const half_t* in; int B, D, H, W;
    const half_t* w; int kpad;        // [Cout][kpad], k = tap * CIN + c, taps in (kz, ky, kx) order
    const float* bias;
    const half_t* resid;
    half_t* out; int Cout;
    int relu;
    int tiles_n, tz, ty, tx;          // tiles per dimension
    int nblocks;
    const half_t* in2;                // CIN2 > 0: second source [B][D][H][W][CIN2] (the block input of a projection shortcut), K columns 27 * CIN ..
};

constexpr int HTZ = 4, HTY = 4, HTX = 8, HHY = HTY + 2, HHX = HTX + 2, HROWS = (HTZ + 2) * HHY * HHX;

// NW waves per workgroup: 4 (2 x 2 or 4 x 1 wave grid, 64 x 32 / 32 x 32 wave tiles; what the launcher uses) or, for
// C_out = 64, 2 waves of 64 x 64 (8 fragment reads per 16 MFMAs instead of 6 per 8, but one wave per SIMD: measured
// slower, kept as a template parameter only).
// TY = 4: 4 x 4 x 8 output voxels (128 rows) per workgroup of four waves.  TY = 8: 4 x 8 x 8 voxels (256 rows) per workgroup of
// eight waves with the same wave tiles: the weights of a tap (and a smaller share of halo) are filled once per 256 rows.
// Used for 32 -> 32, where two such workgroups still fit a CU (64 KB of LDS each): 77 -> 69 us.  At C_in = 64 (94-110 KB,
// one workgroup per CU) it measured equal or slower, also as a persistent kernel that prefetches its next halo.
// CIN2 = 32: one more 32-deep k step behind the 27 taps -- the rows of a SECOND tensor (the residual block's input x, 32 channels) against
// weight columns 27 * CIN .. + 32: ResidualBlock3D's 1x1x1 projection shortcut summed into conv2's accumulators (pcd_conv3d_desc_t.in2).
// x's 128 rows (8 KB) and those 32 weight columns (4 KB) are requested with the halo, wait in registers and pass through the weight ring
// after the last tap (no LDS of their own: two workgroups must still fit a CU).
template <int CIN, int BN, int G, int NSTAGE, int NW, int TY = 4, int CIN2 = 0>
__global__ __launch_bounds__(64 * NW) void conv3d_halo_kernel(HaloParams p) {
    constexpr int NT = 64 * NW;
    constexpr int HHY = TY + 2, HROWS = (HTZ + 2) * HHY * HHX, ROWS = HTZ * TY * HTX, YSH = TY == 8 ? 3 : 2;   // shadow the 128-row geometry
    constexpr int RB = CIN * 2, P = RB, CPR = RB / 16, KS = CIN / 32;
    constexpr int NXV = CIN == 64 ? 3 : 1;                      // address variants per k_x (the swizzle of C_in = 64 depends on hx)
    constexpr int HALO_BYTES = HROWS * P;
    constexpr int BST = G * BN * RB;                          // bytes per weight stage
    constexpr int WAVES_N = (NW >= 4) ? BN / 32 : 1, WAVES_M = NW / WAVES_N, WMR = ROWS / WAVES_M, MI = WMR / 16;
    constexpr int WNC = BN / WAVES_N, NI = WNC / 16;
    constexpr int OUT_LD = BN * 2 + 16;
    constexpr int NS = 27 / G;
    constexpr int HIT = (HROWS * CPR + NT - 1) / NT;          // halo chunks per thread
    constexpr int RPI = 1024 / RB;                            // weight rows per wave-wide DMA instruction
    constexpr int NINSTR = G * BN / RPI;
    constexpr int U = (NINSTR + NW - 1) / NW;                       // DMA instructions per wave and stage (uniform, so
    constexpr int DUMP = (NINSTR % NW) ? 1024 : 0;             // the vmcnt arithmetic is: spare ones hit a dump KB)
    static_assert(27 % G == 0 && ROWS * OUT_LD <= HALO_BYTES && HALO_BYTES % 16 == 0 && NS >= NSTAGE, "layout");
    // second source: its rows and weight columns take over the weight ring once the 27 taps are done (two workgroups per CU need <= 80 KB each)
    constexpr int X2_OFF = HALO_BYTES, W2_OFF = X2_OFF + ROWS * 64;
    static_assert(CIN2 == 0 || (CIN2 == 32 && ROWS * 4 % NT == 0 && BN * 4 <= NT && (ROWS + BN) * 64 <= NSTAGE * BST),
                  "second source: 64-byte rows, one k step, inside the weight ring");
    __shared__ __attribute__((aligned(16))) char smem[HALO_BYTES + NSTAGE * BST + DUMP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // consecutive workgroups go round-robin over the 8 XCDs: give each XCD a contiguous run of tiles so that
    // neighbouring tiles (which share halo voxels) meet in the same L2
    int bid = blockIdx.x;
    if ((p.nblocks & 7) == 0) bid = (bid & 7) * (p.nblocks >> 3) + (bid >> 3);
    const int tn = bid % p.tiles_n; int t = bid / p.tiles_n;
    const int tx = t % p.tx; t /= p.tx;
    const int ty = t % p.ty; t /= p.ty;
    const int tz = t % p.tz; const int b = t / p.tz;
    const int z0 = tz * HTZ, y0 = ty * TY, x0 = tx * HTX, n0 = tn * BN;

    // ---- halo: global -> registers -> LDS (once)
    // (loads are unconditional from a clamped address and zeroed afterwards: no divergent branches,
    //  all HIT loads in flight together)
    half8 hv[HIT];
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int c = it * NT + tid;
        const int row = c / CPR, ch = c - row * CPR;
        const int hx = row % HHX; const int r2 = row / HHX;
        const int hy = r2 % HHY, hz = r2 / HHY;
        const int iz = z0 - 1 + hz, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = row < HROWS && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H &&
                        (unsigned)ix < (unsigned)p.W;
        okmask |= ok ? 1u << it : 0u;
        const int cz = min(max(iz, 0), p.D - 1), cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
        hv[it] = *(const half8*)(p.in + ((((int64_t)b * p.D + cz) * p.H + cy) * p.W + cx) * CIN + ch * 8);
    }

    // rows of the second source and its 32 weight columns: requested with the halo, stored behind it
    constexpr int X2IT = CIN2 ? ROWS * 4 / NT : 1;
    half8 x2v[X2IT], w2v;
    if constexpr (CIN2 > 0) {
#pragma unroll
        for (int it = 0; it < X2IT; ++it) {
            const int c = it * NT + tid, m = c >> 2, ch = c & 3;
            const int x = m & 7, y = (m >> 3) & (TY - 1), z = m >> (3 + YSH);
            x2v[it] = *(const half8*)(p.in2 + ((((int64_t)b * p.D + z0 + z) * p.H + y0 + y) * p.W + x0 + x) * CIN2 + ch * 8);
        }
        int nn = n0 + (tid >> 2);
        nn = nn < p.Cout ? nn : p.Cout - 1;
        w2v = *(const half8*)(p.w + (int64_t)nn * p.kpad + 27 * CIN + (tid & 3) * 8);
    }

    auto stageB = [&](int s, int buf) {
        char* base = smem + HALO_BYTES + buf * BST;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = wave + NW * u;
            const bool real = it < NINSTR;
            const int row = (real ? it : 0) * RPI + lane / CPR;
            const int g = row / BN, n = row % BN;
            const int lch = (lane % CPR) ^ (RB == 128 ? (n >> 1) & 7 : (-(n >> 2)) & 3);
            int nn = n0 + n;
            nn = nn < p.Cout ? nn : p.Cout - 1;
            cglds16_asm(p.w + (int64_t)nn * p.kpad + (s * G + g) * CIN + lch * 8,
                        real ? base + it * 1024 : smem + HALO_BYTES + NSTAGE * BST);
        }
    };
    stageB(0, 0);
    if (NS > 1) stageB(1, 1);
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int c = it * NT + tid;
        const int row = c / CPR, ch = c - row * CPR;
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const int hx = row % HHX, hy = (row / HHX) % HHY;
        const int sw = CIN == 64 ? (((hx >> 1) & 1) << 1) | ((hy & 1) << 2) : (hy & 1) << 1;
        if (row < HROWS) *(half8*)(smem + row * P + ((ch ^ sw) << 4)) = (okmask >> it) & 1 ? hv[it] : zero8;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4;
    // A fragment address of row block i for a tap (kz, ky, kx) and k step ks: areg[i][kx][ks ^ (ky & 1)] + the tap's
    // voxel offset (a compile-time immediate); the low bits carry the swizzle of the voxel the tap lands on
    int areg[MI][NXV][2], boff[NI][KS];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * WMR + i * 16 + (lane & 15);
        const int x = m & 7, y = (m >> 3) & (TY - 1), z = m >> (3 + YSH);
        const int vrow = ((z * HHY + y) * HHX + x) * P;
#pragma unroll
        for (int kx = 0; kx < NXV; ++kx)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int hyb = (e ^ y) & 1;                                     // (ks ^ hy) & 1
                areg[i][kx][e] = vrow + (CIN == 64 ? (hyb << 6) | ((q ^ ((((x + kx) >> 1) & 1) << 1)) << 4)
                                                   : (q ^ (hyb << 1)) << 4);
            }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = wn * WNC + j * 16 + (lane & 15);
        const int sw = RB == 128 ? (n >> 1) & 7 : (-(n >> 2)) & 3;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) boff[j][ks] = HALO_BYTES + n * RB + (((ks * 4 + q) ^ sw) << 4);
    }

    // Fragment registers are double buffered per 32-wide k step: while the MFMAs of step u run, the fragments
    // of step u+1 are on their way from LDS (A from the halo, which never changes; B from the weight stage).
    // One step ahead, not one tap: a wave can have at most 15 LDS reads outstanding (lgkmcnt is 4 bits).
    half8 af[2][MI], bf[2][NI];
    auto readA = [&](int par, int tap, int ks) {
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        const int voff = ((kz * HHY + ky) * HHX + kx) * P;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            af[par][i] = *(const half8*)(smem + areg[i][NXV == 3 ? kx : 0][(ks ^ ky) & 1] + voff);
    };
    auto readB = [&](int par, const char* bbuf, int g, int ks) {
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[par][j] = *(const half8*)(bbuf + boff[j][ks] + g * BN * RB);
    };
    // Weight stages: buffer s % NSTAGE.  The barrier of stage s publishes stage s+1 (every wave waited for its own
    // DMAs of it; with four buffers the DMAs of stage s+2 stay in flight behind a counted vmcnt) and frees the buffer
    // of stage s-1 for stage s+NSTAGE-1.  A stage is therefore requested NSTAGE-2 taps before the barrier that needs
    // it: one tap of MFMAs (~500 cycles per wave) does not cover an L2 round trip, two do.  The fragment prefetch runs
    // one k step ahead, also across a stage boundary.  Fully unrolled (27 taps): with a loop back-edge the compiler's
    // waitcnt pass falls back to lgkmcnt(0) in front of the MFMAs, which drains the prefetch it is meant to overlap.
    static_assert(NSTAGE == 3 || NSTAGE == 4, "the prefetch protocol below is written for three or four weight buffers");
    if (NSTAGE == 4 && NS > 2) {
        stageB(2, 2);
        conv_wait_vmcnt<2 * U>();
    } else {
        conv_wait_vmcnt<(NS > 1) ? U : 0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    readA(0, 0, 0);
    readB(0, smem, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) {
            if (NSTAGE == 4 && s + 2 < NS) conv_wait_vmcnt<U>(); else conv_wait_vmcnt<0>();
            conv_reads_landed();
            __builtin_amdgcn_s_barrier();   // raw: __syncthreads() adds its own waits
            if (s + NSTAGE - 1 < NS) stageB(s + NSTAGE - 1, (s + NSTAGE - 1) % NSTAGE);
        }
        const char* bcur = smem + (s % NSTAGE) * BST;
        const char* bnxt = smem + ((s + 1) % NSTAGE) * BST;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int par = ((s * G + g) * KS + ks) & 1;
                if (ks + 1 < KS) {
                    readA(par ^ 1, s * G + g, ks + 1);
                    readB(par ^ 1, bcur, g, ks + 1);
                } else if (g + 1 < G) {
                    readA(par ^ 1, s * G + g + 1, 0);
                    readB(par ^ 1, bcur, g + 1, 0);
                } else if (s + 1 < NS) {
                    readA(par ^ 1, (s + 1) * G, 0);
                    readB(par ^ 1, bnxt, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[par][i], bf[par][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }

    if constexpr (CIN2 > 0) {
        // the 28th k step: second-source rows x the shortcut's weight columns, through the weight ring (every tap has been read and no
        // DMA is in flight: the last stage was waited for two taps ago).  64-byte rows, chunk c of row r at c ^ ((-(r >> 2)) & 3) (swz<32>)
        __syncthreads();
#pragma unroll
        for (int it = 0; it < X2IT; ++it) {
            const int c = it * NT + tid, m = c >> 2, ch = c & 3;
            *(half8*)(smem + X2_OFF + m * 64 + ((ch ^ ((-(m >> 2)) & 3)) << 4)) = x2v[it];
        }
        if (tid < BN * 4) *(half8*)(smem + W2_OFF + (tid >> 2) * 64 + (((tid & 3) ^ ((-(tid >> 4)) & 3)) << 4)) = w2v;
        __syncthreads();
        half8 a2[MI], b2[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wm * WMR + i * 16 + (lane & 15);
            a2[i] = *(const half8*)(smem + X2_OFF + m * 64 + ((q ^ ((-(m >> 2)) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = wn * WNC + j * 16 + (lane & 15);
            b2[j] = *(const half8*)(smem + W2_OFF + n * 64 + ((q ^ ((-(n >> 2)) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[i], b2[j], acc[i][j], 0, 0, 0);
    }

    // epilogue (same rounding points as conv3d_igemm_kernel): bias -> fp16 in LDS -> 16-B row stores
    const int colq = lane & 15;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int lcol = wn * WNC + j * 16 + colq;
        const float bc = (p.bias != nullptr && n0 + lcol < p.Cout) ? p.bias[n0 + lcol] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = wm * WMR + i * 16 + q * 4 + r;
                float v = acc[i][j][r] + bc;
                if (p.relu && p.resid == nullptr) v = fmaxf(v, 0.f);
                *(half_t*)(smem + lrow * OUT_LD + lcol * 2) = to_half_sat(v);
            }
    }
    __syncthreads();
    constexpr int OCPR = BN / 8, TOTAL = ROWS * OCPR;
#pragma unroll
    for (int it = 0; it < TOTAL / NT; ++it) {
        const int idx = it * NT + tid;
        const int lrow = idx / OCPR, ch = idx - lrow * OCPR;
        const int col = n0 + ch * 8;
        if (col < p.Cout) {
            const int x = lrow & 7, y = (lrow >> 3) & (TY - 1), z = lrow >> (3 + YSH);
            const int64_t orow = (((int64_t)b * p.D + z0 + z) * p.H + y0 + y) * p.W + x0 + x;
            half8 v = *(const half8*)(smem + lrow * OUT_LD + ch * 16);
            if (p.resid != nullptr) {
                const half8 rs = *(const half8*)(p.resid + orow * p.Cout + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[e] + (float)rs[e];
                    if (p.relu) f = fmaxf(f, 0.f);
                    v[e] = to_half_sat(f);
                }
            }
            *(half8*)(p.out + orow * p.Cout + col) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d(k4, s2, p1) + ReLU, C_in = 128 -> C_out = 64 (decoder.6, reference networks.py:2253), with the input halo in LDS.
// As eight output-parity classes through the implicit GEMM above this layer is a 128 x 64 tile per class (43 FLOP per gathered
// byte) and every input voxel is fetched 8 x 8 times: 556 TFLOP/s.  Here a workgroup owns 4 x 4 x 8 INPUT voxels, loads their
// 6 x 6 x 10 halo (offsets -1 .. +1 cover every class: o = 2 i - 1 + k) ONCE -- 92 KB -- and computes all eight classes
// (8 x 128 output voxels x 64 channels) from it; only the weights stream (1 MB per workgroup, 16-KB stages in a ring of four).
//   * eight waves: two classes at a time (px = 0 / 1 of one (pz, py) pair), four waves per class as 2 voxel halves x 2 channel
//     halves, wave tile 64 voxels x 32 channels;
//   * the product is taken TRANSPOSED, D[channel][voxel] (weights = MFMA A operand, voxels = B), so a lane's accumulators are
//     four consecutive channels of one voxel: two 16-channel blocks trade halves (v_permlane16_swap) and every lane stores
//     16 contiguous bytes of an output row -- no LDS staging, so the weight ring keeps streaming through the epilogues;
//   * halo image: voxel v = (hz * 6 + hy) * 10 + hx at v * 256 B (one 64-bank row per voxel), its 16-byte chunk c at slot
//     c ^ ((hx & 7) << 1): a ds_read_b128 lane group holds 8 voxels with chunk c and 8 with chunk c ^ 1 (MI355X_MICROARCH.md, LDS),
//     the 16 voxels of an MFMA block have 8 distinct x twice, so the slots are 16 distinct ones for every tap shift;
//   * weight stage = one tap x one 64-channel half of K x two classes (2 x 64 rows of 128 B, the swz<64> of the GEMM),
//     requested three stages ahead, counted vmcnt (the epilogue's four stores per wave are counted too).
struct ConvTHaloParams {
    const half_t* in; int B, D, H, W;
    const half_t* w[8];               // class 4 pz + 2 py + px: [64][8 * 128], tap t = (tz * 2 + ty) * 2 + tx reads input offset p - t per axis
    const float* bias;
    half_t* out;                      // [B][2D][2H][2W][64]
    int tz, ty, tx, nblocks;
};

__global__ __launch_bounds__(512) void convT3d_halo_kernel(ConvTHaloParams p) {
    constexpr int CIN = 128, COUT = 64, VB = CIN * 2, NW = 8, NT = 512, NSTAGE = 4;
    constexpr int HALO_BYTES = HROWS * VB;                   // 360 voxels x 256 B
    constexpr int BST = 2 * COUT * 128;                      // stage: 2 classes x 64 rows x 64 k
    constexpr int U = BST / 1024 / NW;                       // DMA instructions per wave and stage (2)
    constexpr int NSTG = 64;                                 // 4 class pairs x 8 taps x 2 K halves
    constexpr int HIT = (HROWS * 16 + NT - 1) / NT;
    constexpr int NSTORE = 4;                                // epilogue stores per wave and class pair
    static_assert(U * NW * 1024 == BST && HALO_BYTES + NSTAGE * BST <= 160 * 1024, "layout");
    __shared__ __attribute__((aligned(16))) char smem[HALO_BYTES + NSTAGE * BST];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave >> 2, wv = (wave >> 1) & 1, wc = wave & 1;   // class in the pair (= px), voxel half, channel half
    int bid = blockIdx.x;
    if ((p.nblocks & 7) == 0) bid = (bid & 7) * (p.nblocks >> 3) + (bid >> 3);
    int t = bid;
    const int tx = t % p.tx; t /= p.tx;
    const int ty = t % p.ty; t /= p.ty;
    const int tz = t % p.tz; const int b = t / p.tz;
    const int z0 = tz * HTZ, y0 = ty * HTY, x0 = tx * HTX;

    // ---- halo: global -> registers (all loads in flight) -> LDS
    half8 hv[HIT];
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int c = it * NT + tid;
        const int row = c >> 4, ch = c & 15;
        const int hx = row % HHX; const int r2 = row / HHX;
        const int hy = r2 % HHY, hz = r2 / HHY;
        const int iz = z0 - 1 + hz, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = row < HROWS && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        okmask |= ok ? 1u << it : 0u;
        const int cz = min(max(iz, 0), p.D - 1), cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
        hv[it] = *(const half8*)(p.in + ((((int64_t)b * p.D + cz) * p.H + cy) * p.W + cx) * CIN + ch * 8);
    }

    const int q = lane >> 4, n16 = lane & 15;
    float bv[2][4];                                    // bias of this lane's channels (requested before any DMA: older in vmcnt order)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = p.bias != nullptr ? p.bias[wc * 32 + i * 16 + q * 4 + r] : 0.f;

    // stage S = (pair, tap, K half): instruction it = wave + 8 u covers rows it * 8 .. + 7 of the 128 (class u, channels (wave * 8 ..) + lane / 8)
    auto stageW = [&](int S, int buf) {
        const int pair = S >> 4, tap = (S >> 1) & 7, kh = S & 1;
        char* base = smem + HALO_BYTES + buf * BST;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = wave * 8 + (lane >> 3);
            const int lch = (lane & 7) ^ ((n >> 1) & 7);
            cglds16_asm(p.w[2 * pair + u] + n * (8 * CIN) + tap * CIN + kh * 64 + lch * 8, base + (wave + NW * u) * 1024);
        }
    };
    stageW(0, 0);
    stageW(1, 1);
    stageW(2, 2);
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int c = it * NT + tid;
        const int row = c >> 4, ch = c & 15;
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const int hx = row % HHX;
        if (row < HROWS) *(half8*)(smem + row * VB + ((ch ^ ((hx & 7) << 1)) << 4)) = (okmask >> it) & 1 ? hv[it] : zero8;
    }

    // voxel blocks of this wave: j -> (z = 2 wv + (j >> 1), y = 2 (j & 1) + (n16 >> 3), x = n16 & 7); halo voxel without a tap = + (1, 1, 1)
    const int vx = n16 & 7;
    int vrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vrow[j] = (((2 * wv + (j >> 1) + 1) * HHY + 2 * (j & 1) + (n16 >> 3) + 1) * HHX + vx + 1) * VB;
    // weight rows of this wave in a stage: class cw, channels wc * 32 + i * 16 + n16
    int wrow[2], wsw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = wc * 32 + i * 16 + n16;
        wrow[i] = HALO_BYTES + (cw * COUT + n) * 128;
        wsw[i] = (n >> 1) & 7;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    conv_wait_vmcnt<2 * U>();                          // stage 0 (and, before it, the halo loads) landed; stages 1, 2 in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int OD = 2 * p.D, OH = 2 * p.H, OW = 2 * p.W;
#pragma unroll 1
    for (int pair = 0; pair < 4; ++pair) {
        const int pz = pair >> 1, py = pair & 1, px = cw;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int S = pair * 16 + st;
            if (S + 1 < NSTG) {
                // publish stage S + 1: behind it at most stage S + 2 (and, right after a pair's epilogue, that epilogue's stores) may stay in flight
                if (st < 2) {
                    if (pair > 0) conv_wait_vmcnt<U + NSTORE>(); else conv_wait_vmcnt<U>();
                } else if (st == 14 && pair == 3) {
                    conv_wait_vmcnt<0>();
                } else {
                    conv_wait_vmcnt<U>();
                }
                conv_reads_landed();
                __builtin_amdgcn_s_barrier();
                if (S + 3 < NSTG) stageW(S + 3, (S + 3) & 3);
            }
            const int tap = st >> 1, kh = st & 1;
            const int dz = pz - (tap >> 2), dy = py - ((tap >> 1) & 1), dx = px - (tap & 1);
            const int voff = ((dz * HHY + dy) * HHX + dx) * VB;
            const int sw = ((vx + 1 + dx) & 7) << 1;
            const char* wb = smem + (S & 3) * BST;
#pragma unroll
            for (int ksl = 0; ksl < 2; ++ksl) {
                half8 af[2], bf[4];
                const int kc = (kh * 2 + ksl) * 4 + q;                     // 16-byte chunk of the voxel's 128 channels
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *(const half8*)(wb + wrow[i] + (((ksl * 4 + q) ^ wsw[i]) << 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(smem + vrow[j] + voff + ((kc ^ sw) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        // epilogue of the pair: bias, ReLU, fp16; channel blocks 0 / 1 trade halves, lane group q stores 8 consecutive channels of
        // block (q & 1) at offset 8 * (q >> 1) of its voxel's output row
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned pk[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                half2_t lo2, hi2;
                lo2[0] = to_half_sat(fmaxf(acc[i][j][0] + bv[i][0], 0.f)); lo2[1] = to_half_sat(fmaxf(acc[i][j][1] + bv[i][1], 0.f));
                hi2[0] = to_half_sat(fmaxf(acc[i][j][2] + bv[i][2], 0.f)); hi2[1] = to_half_sat(fmaxf(acc[i][j][3] + bv[i][3], 0.f));
                pk[i][0] = __builtin_bit_cast(unsigned, lo2);
                pk[i][1] = __builtin_bit_cast(unsigned, hi2);
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            const int z = 2 * wv + (j >> 1), y = 2 * (j & 1) + (n16 >> 3);
            const int64_t orow = (((int64_t)b * OD + 2 * (z0 + z) + pz) * OH + 2 * (y0 + y) + py) * OW + 2 * (x0 + vx) + px;
            *(u32x4*)(p.out + orow * COUT + wc * 32 + (q & 1) * 16 + (q >> 1) * 8) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Conv3d(k4, s2, p1) + ReLU, C_in = 64 -> C_out = 64 (encoder.3, reference networks.py:2229), with LDS-resident input.
// Through the implicit GEMM this layer is a 128 x 64 tile at 43 FLOP per gathered byte (every input voxel fetched 8 times, every
// tile re-gathering its 64 taps): 584 TFLOP/s, the slowest large layer of the encoder.  The stride-2 kernel splits by INPUT parity:
// out(o) = sum_k in(2 o - 1 + k) w(k), and per axis the even inputs are reached by k = 1, 3 at sub-grid index o, o + 1, the odd ones
// by k = 0, 2 at o - 1, o -- eight classes, each a 2 x 2 x 2 stride-1 convolution of one sub-sampled grid, all summed into the SAME
// output.  A workgroup owns 4 x 4 x 8 output voxels; per class it holds the 5 x 5 x 9 sub-grid halo (input voxel 2 (o0 + h) - parity,
// in the 6 x 10 pitch and with the swizzle of conv3d_halo_kernel's C_in = 64 image, so the same conflict-free fragment reads)
// and runs the class's 8 taps from it; the next class's halo is requested at the class's first tap (into registers) and swapped in
// behind one barrier; the weights stream through the same ring of four 8-KB stages over all 64 (class, tap) stages; the product
// is taken transposed with 16-byte direct stores like convT3d_halo_kernel.  70 KB of LDS: two workgroups per CU, the other one
// computes through this one's halo swaps.
struct ConvS2HaloParams {
    const half_t* in; int B, D, H, W;    // input grid (even dims); output grid D/2 x H/2 x W/2
    const half_t* w; int kpad;            // [64][kpad], k = ((kz * 4 + ky) * 4 + kx) * 64 + c
    const float* bias;
    half_t* out;
    int relu;
    int tz, ty, tx, nblocks;
};

__global__ __launch_bounds__(256) void conv3d_k4s2_halo_kernel(ConvS2HaloParams p) {
    constexpr int CIN = 64, COUT = 64, P = CIN * 2, NW = 4, NT = 256, NSTAGE = 4;
    constexpr int HALO_BYTES = HROWS * P;                    // the 6 x 6 x 10 pitch of the stride-1 kernels (5 x 5 x 9 voxels used)
    constexpr int BST = COUT * P;                            // one tap: 64 rows x 128 B
    constexpr int U = BST / 1024 / NW;                       // DMA instructions per wave and stage (2)
    constexpr int NSTG = 64;                                 // 8 classes x 8 taps
    constexpr int SV = 5 * 5 * 9, HIT = (SV * 8 + NT - 1) / NT;   // sub-halo voxels; 16-byte chunks per thread (8)
    __shared__ __attribute__((aligned(16))) char smem[HALO_BYTES + NSTAGE * BST];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // voxel half (64 of the 128), channel half
    int bid = blockIdx.x;
    if ((p.nblocks & 7) == 0) bid = (bid & 7) * (p.nblocks >> 3) + (bid >> 3);
    int t = bid;
    const int tx = t % p.tx; t /= p.tx;
    const int ty = t % p.ty; t /= p.ty;
    const int tz = t % p.tz; const int b = t / p.tz;
    const int z0 = tz * HTZ, y0 = ty * HTY, x0 = tx * HTX;   // output block origin

    // this thread's chunks of a class's sub-halo: chunk c = it * 256 + tid -> voxel c >> 3 = (hz * 5 + hy) * 9 + hx, 16-byte piece c & 7.
    // Everything that does not depend on the class is worked out once: the chunk's element offset for class (0, 0, 0) (input voxel 2 (o0 + h)),
    // its LDS address, and six validity bits (per axis: index 2 (o0 + h) inside the grid for an even class / 2 (o0 + h) - 1 for an odd one);
    // class (cz, cy, cx) then reads (cz H W + cy W + cx) C_in elements earlier
    half8 hv[HIT];
    int hoff[HIT], hlds[HIT];
    unsigned hflags = 0;                                     // 6 bits per chunk... packed 4 chunks per word would not fit: one word per axis pair below
    unsigned fz = 0, fy = 0, fx = 0;                         // bit 2 it: even class valid, bit 2 it + 1: odd class valid
    (void)hflags;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int c = it * NT + tid;
        const int v = c >> 3, ch = c & 7;
        const int hx = v % 9; const int r2 = v / 9;
        const int hy = r2 % 5, hz = r2 / 5;
        const int iz = 2 * (z0 + hz), iy = 2 * (y0 + hy), ix = 2 * (x0 + hx);
        const bool real = v < SV;
        fz |= ((real && iz < p.D) ? 1u : 0u) << (2 * it) | ((real && iz >= 1 && iz - 1 < p.D) ? 2u : 0u) << (2 * it);
        fy |= ((iy < p.H) ? 1u : 0u) << (2 * it) | ((iy >= 1 && iy - 1 < p.H) ? 2u : 0u) << (2 * it);
        fx |= ((ix < p.W) ? 1u : 0u) << (2 * it) | ((ix >= 1 && ix - 1 < p.W) ? 2u : 0u) << (2 * it);
        hoff[it] = (((b * p.D + iz) * p.H + iy) * p.W + ix) * CIN + ch * 8;
        const int sw = (((hx >> 1) & 1) << 1) | ((hy & 1) << 2);
        hlds[it] = real ? ((hz * HHY + hy) * HHX + hx) * P + ((ch ^ sw) << 4) : -1;
    }
    unsigned okmask = 0;
    auto halo_request = [&](int cls) __attribute__((always_inline)) {
        const int cz = cls >> 2, cy = (cls >> 1) & 1, cx = cls & 1;
        const int delta = ((cz * p.H + cy) * p.W + cx) * CIN;
        const unsigned m = (fz >> cz) & (fy >> cy) & (fx >> cx);
        okmask = 0;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const bool ok = (m >> (2 * it)) & 1;
            okmask |= ok ? 1u << it : 0u;
            const half_t* g = p.in + (ok ? hoff[it] - delta : 0);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hv[it]) : "v"(g) : "memory");      // (the kernel owns every vmcnt wait)
        }
    };
    auto halo_store = [&]() __attribute__((always_inline)) {
        // the loads above are asm: to the compiler their registers were defined at the request.  Re-define them HERE, behind the wait that retired
        // the loads, so that nothing computed from them (the zero-fill select below) can be scheduled ahead of their landing
#pragma unroll
        for (int it = 0; it < HIT; ++it) asm volatile("" : "+v"(hv[it]));
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
            if (hlds[it] >= 0) *(half8*)(smem + hlds[it]) = (okmask >> it) & 1 ? hv[it] : zero8;
        }
    };
    // stage S = class * 8 + tap (tz, ty, tx): the k4 tap (2 t + 1 - parity) per axis; rows wave * 16 + 8 u + lane / 8 of the 64
    auto stageW = [&](int S, int buf) __attribute__((always_inline)) {
        const int cls = S >> 3, st = S & 7;
        const int kz = 2 * (st >> 2) + 1 - (cls >> 2), ky = 2 * ((st >> 1) & 1) + 1 - ((cls >> 1) & 1), kx = 2 * (st & 1) + 1 - (cls & 1);
        const int t4 = (kz * 4 + ky) * 4 + kx;
        char* base = smem + HALO_BYTES + buf * BST;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = (wave * U + u) * 8 + (lane >> 3);
            const int lch = (lane & 7) ^ ((n >> 1) & 7);
            cglds16_asm(p.w + (int64_t)n * p.kpad + t4 * CIN + lch * 8, base + (wave * U + u) * 1024);
        }
    };

    const int q = lane >> 4, n16 = lane & 15;
    float bv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = p.bias != nullptr ? p.bias[wn * 32 + i * 16 + q * 4 + r] : 0.f;
    halo_request(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    halo_store();
    stageW(0, 0);
    stageW(1, 1);
    stageW(2, 2);

    // voxel blocks of this wave: j -> local (z = 2 wm + (j >> 1), y = 2 (j & 1) + (n16 >> 3), x = n16 & 7); halo voxel of tap t = + t
    const int vx = n16 & 7, vyp = n16 >> 3;
    int vrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vrow[j] = (((2 * wm + (j >> 1)) * HHY + 2 * (j & 1) + vyp) * HHX + vx) * P;
    int wrow[2], wsw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = wn * 32 + i * 16 + n16;
        wrow[i] = HALO_BYTES + n * P;
        wsw[i] = (n >> 1) & 7;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    conv_wait_vmcnt<2 * U>();                          // stage 0 landed; stages 1, 2 in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int cls = 0; cls < 8; ++cls) {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int S = cls * 8 + st;
            if (S + 1 < NSTG) {
                // publish stage S + 1: behind it stage S + 2 may stay in flight -- and, at a class's taps 1 and 2, the next class's eight halo loads,
                // issued behind stage S + 3 at tap 0
                if ((st == 1 || st == 2) && cls < 7) conv_wait_vmcnt<U + HIT>();
                else if (S + 2 < NSTG) conv_wait_vmcnt<U>();
                else conv_wait_vmcnt<0>();
                conv_reads_landed();
                __builtin_amdgcn_s_barrier();
                if (S + 3 < NSTG) stageW(S + 3, (S + 3) & 3);
            }
            if (st == 0 && cls < 7) halo_request(cls + 1);
            const int voff = (((st >> 2) * HHY + ((st >> 1) & 1)) * HHX + (st & 1)) * P;
            const int sw = ((((vx + (st & 1)) >> 1) & 1) << 1) | (((vyp + ((st >> 1) & 1)) & 1) << 2);
            const char* wb = smem + (S & 3) * BST;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 af[2], bf[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *(const half8*)(wb + wrow[i] + (((ks * 4 + q) ^ wsw[i]) << 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(smem + vrow[j] + voff + (((ks * 4 + q) ^ sw) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (cls < 7) {
            // swap the halo: every wave has read the last tap's fragments; the next class's chunks landed long ago (tap 3's wait retired them)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            halo_store();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (published by the next tap's barrier)
        }
    }
    // epilogue: bias, ReLU, fp16; channel blocks 0 / 1 trade halves, lane group q stores 8 consecutive channels of block (q & 1) at offset 8 (q >> 1)
    const int OD = p.D >> 1, OH = p.H >> 1, OW = p.W >> 1;
    const float lo = p.relu ? 0.f : -65504.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned pk[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
            half2_t lo2, hi2;
            lo2[0] = (half_t)__builtin_amdgcn_fmed3f(acc[i][j][0] + bv[i][0], lo, 65504.f); lo2[1] = (half_t)__builtin_amdgcn_fmed3f(acc[i][j][1] + bv[i][1], lo, 65504.f);
            hi2[0] = (half_t)__builtin_amdgcn_fmed3f(acc[i][j][2] + bv[i][2], lo, 65504.f); hi2[1] = (half_t)__builtin_amdgcn_fmed3f(acc[i][j][3] + bv[i][3], lo, 65504.f);
            pk[i][0] = __builtin_bit_cast(unsigned, lo2);
            pk[i][1] = __builtin_bit_cast(unsigned, hi2);
        }
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        const int z = 2 * wm + (j >> 1), y = 2 * (j & 1) + vyp;
        const int64_t orow = (((int64_t)b * OD + z0 + z) * OH + y0 + y) * OW + x0 + vx;
        *(u32x4*)(p.out + orow * COUT + wn * 32 + (q & 1) * 16 + (q >> 1) * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// k3 / s1 / p1, C_in = 64, with the WEIGHTS IN REGISTERS: conv3d_halo_kernel above reads, per 32-deep k step and wave, four voxel
// fragments and two weight fragments from LDS for eight MFMAs -- with eight waves per CU the LDS pipe is busy 384 cycles per 256 MFMA
// cycles, which caps the matrix pipe at two thirds (and it streams every tap's weights through an LDS ring behind a barrier per tap).
// Here a workgroup owns 4 x 8 x 8 output voxels (256 rows; halo 6 x 10 x 10 = 76.8 KB: two workgroups per CU, nothing else in LDS), a
// wave 128 voxels x 32 channels, and the weight fragments never touch LDS: they are stored once per model in MFMA-fragment order
// (pcd_conv3d_pack_wfrag: [tile][tap][channel half][k step][16-channel block][lane][8]) so that a wave's fragment is ONE coalesced
// 1-KB global load straight into the registers the MFMA reads, two taps ahead.  Per k step and wave: eight voxel fragment reads (LDS)
// + two weight loads (L1/L2: 32 B per cycle and CU, half the L1 rate) for 16 MFMAs -- the LDS pipe is busy 512 cycles per 512 MFMA
// cycles at two waves per SIMD -- and there is NO barrier after the halo is in place: the 27 taps are straight-line code that the
// compiler pipelines (counted vmcnt / lgkmcnt of its own).  Transposed product, 16-byte direct stores (+ residual), as convT3d_halo_kernel.
struct HaloWregParams {
    const half_t* in; int B, D, H, W;
    const half_t* in2; int cin2;      // optional second source (pcd_conv3d_desc_t.in2, 32 or 64 channels): one or two more 32-deep k steps behind the taps, its
                                      // voxel fragments straight from global memory (a lane's 16 bytes = 8 of a row's channels), its weights = the stage
                                      // behind the last tap's
    const half_t* wfrag;              // [tiles_n][27 SPT + 1 stages][NWC][KSS][2][64][8]: a stage = up to 64 channels of one tap (SPT per tap), + the second source's
    const float* bias;
    const half_t* resid;
    half_t* out; int Cout;
    int relu;
    int tiles_n, tz, ty, tx, nblocks;
};

// CIN = 64 or 32 (32: one k step per tap, 64-byte voxel rows, 38.4 KB of halo: four workgroups per CU); NWC = wave columns: a tile is 32 NWC channels
// wide and a workgroup NWR x NWC waves (C_out = 32 layers: NWC = 1).  CIN = 128: 256-byte voxel rows (the swizzle of convT3d_halo_kernel), a 153.6-KB halo
// = ONE workgroup per CU, so it runs eight waves (tiles of 128 channels, NWC = 4); a tap is then two weight stages of 64 channels.
// ABL: timing ablations of the <64, 2> instance (pcd_conv3d_config + 32768 x bits; OUTPUTS WRONG): 1 = the halo is not loaded (zeros are written to LDS), 2 = no halo staging
// at all, 4 = no taps (no fragment reads, no MFMAs, no weight loads)
template <int CIN, int NWC, int NWR = 2, int ABL = 0>
__global__ __launch_bounds__(64 * NWR * NWC, 2) void conv3d_halo_wreg_kernel(HaloWregParams p) {      // >= two waves per SIMD: at most 256 registers
    constexpr int P = CIN * 2, CPV = CIN / 8, NT = 64 * NWR * NWC, TY8 = 8, HHY8 = TY8 + 2, TC = 32 * NWC;
    constexpr int SPT = CIN > 64 ? CIN / 64 : 1, KSS = (CIN > 64 ? 64 : CIN) / 32, NSTG = 27 * SPT;      // weight stages per tap, k steps per stage, stages
    constexpr int NB = 16 / NWR;                               // 16-voxel blocks per wave: NWR = 2 wave rows of 128 voxels, or 4 of 64 (C_in 64 -> C_out 32: four waves)
    constexpr int HV = (HTZ + 2) * HHY8 * HHX;                 // 600 halo voxels
    constexpr int HIT = (HV * CPV + NT - 1) / NT;              // 16-byte chunks per thread
    static_assert(HIT <= 32, "one validity bit per chunk in a 32-bit mask");
    __shared__ __attribute__((aligned(16))) char smem[HV * P];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave - wr * NWC;           // voxel rows wr * 16 NB .. of the 256, 32-channel column of the tile
    int bid = blockIdx.x;
    if ((p.nblocks & 7) == 0) bid = (bid & 7) * (p.nblocks >> 3) + (bid >> 3);
    const int tn = bid % p.tiles_n; int t = bid / p.tiles_n;
    const int tx = t % p.tx; t /= p.tx;
    const int ty = t % p.ty; t /= p.ty;
    const int tz = t % p.tz; const int b = t / p.tz;
    const int z0 = tz * HTZ, y0 = ty * TY8, x0 = tx * HTX;

    // ---- halo: global -> registers (all loads in flight) -> LDS, chunk c of voxel (hz, hy, hx) at slot c ^ s(hx, hy) (the C_in = 64 image above)
    if constexpr (!(ABL & 2)) {
        half8 hv[HIT];
        unsigned okmask = 0;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int c = it * NT + tid;
            const int row = c / CPV, ch = c - row * CPV;
            const int hx = row % HHX; const int r2 = row / HHX;
            const int hy = r2 % HHY8, hz = r2 / HHY8;
            const int iz = z0 - 1 + hz, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = row < HV && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            okmask |= ok ? 1u << it : 0u;
            const int cz = min(max(iz, 0), p.D - 1), cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
            if constexpr ((ABL & 1) != 0) hv[it] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            else hv[it] = *(const half8*)(p.in + ((((int64_t)b * p.D + cz) * p.H + cy) * p.W + cx) * CIN + ch * 8);
        }
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int c = it * NT + tid;
            const int row = c / CPV, ch = c - row * CPV;
            const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
            const int hx = row % HHX, hy = (row / HHX) % HHY8;
            const int sw = CIN == 128 ? (hx & 7) << 1 : (CIN == 64 ? (((hx >> 1) & 1) << 1) | ((hy & 1) << 2) : (hy & 1) << 1);
            if (row < HV) *(half8*)(smem + row * P + ((ch ^ sw) << 4)) = (okmask >> it) & 1 ? hv[it] : zero8;
        }
    }
    const int q = lane >> 4, n16 = lane & 15;
    // this wave's weight fragments: stage g, k step ks, channel block j at wf + (((g * NWC + wc) * KSS + ks) * 2 + j) * 512 halfs (+ lane * 8)
    const half_t* wf = p.wfrag + (int64_t)tn * (NSTG + 1) * NWC * KSS * 2 * 512 + lane * 8;
    float bv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias != nullptr ? p.bias[tn * TC + wc * 32 + j * 16 + q * 4 + r] : 0.f;
    // voxel blocks of this wave: block g = wr * NB + i -> local (z = g >> 2, y = 2 (g & 3) + (n16 >> 3), x = n16 & 7)
    const int vx = n16 & 7, vyp = n16 >> 3;
    int vrow[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) vrow[i] = ((((wr * NB + i) >> 2) * HHY8 + 2 * ((wr * NB + i) & 3) + vyp) * HHX + vx) * P;
    f32x4 acc[2][NB];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 wq[3][KSS][2];                                       // weight fragments of stages g, g + 1, g + 2 (ring of three)
    auto wload = [&](int slot, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KSS; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) wq[slot][ks][j] = *(const half8*)(wf + (((g * NWC + wc) * KSS + ks) * 2 + j) * 512);
    };
    wload(0, 0);
    wload(1, 1);
    __syncthreads();                                           // the halo is in place; no barrier from here on

#pragma unroll
    for (int g = 0; g < ((ABL & 4) ? 0 : NSTG); ++g) {
        if (g + 2 < NSTG) wload((g + 2) % 3, g + 2);
        __builtin_amdgcn_sched_barrier(0);                     // the loads stay HERE, two stages ahead of their use (left alone the scheduler sinks them to it)
        const int tap = g / SPT, kh = g % SPT;                 // kh: which 64 channels of the tap
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        const int voff = ((kz * HHY8 + ky) * HHX + kx) * P;
        const int sw = CIN == 128 ? ((vx + kx) & 7) << 1
                                  : (CIN == 64 ? ((((vx + kx) >> 1) & 1) << 1) | (((vyp + ky) & 1) << 2) : ((vyp + ky) & 1) << 1);
#pragma unroll
        for (int ks = 0; ks < KSS; ++ks) {
            half8 vf[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) vf[i] = *(const half8*)(smem + vrow[i] + voff + ((((kh * KSS + ks) * 4 + q) ^ sw) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < NB; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[g % 3][ks][j], vf[i], acc[j][i], 0, 0, 0);
        }
    }
    if (CIN >= 64 && p.in2 != nullptr) {
        // the second source's k steps (cin2 / 32 of them): weights = the stage behind the last tap's, voxel fragments from global memory (row m of in2,
        // channels 32 ks + 8 q .. + 7)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks * 32 < p.cin2) {
                half8 w2[2], x2[NB];
#pragma unroll
                for (int j = 0; j < 2; ++j) w2[j] = *(const half8*)(wf + (((NSTG * NWC + wc) * KSS + ks) * 2 + j) * 512);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int z = (wr * NB + i) >> 2, y = 2 * ((wr * NB + i) & 3) + vyp;
                    x2[i] = *(const half8*)(p.in2 + ((((int64_t)b * p.D + z0 + z) * p.H + y0 + y) * p.W + x0 + vx) * p.cin2 + ks * 32 + q * 8);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < NB; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[j], x2[i], acc[j][i], 0, 0, 0);
            }
        }
    }
    // epilogue: bias -> fp16 (the rounding point of the other kernels) (+ residual) (+ ReLU); channel blocks 0 / 1 trade halves, lane group q stores
    // 8 consecutive channels of block (q & 1) at offset 8 (q >> 1)
    const float lo = (p.relu && p.resid == nullptr) ? 0.f : -65504.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        unsigned pk[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
            half2_t lo2, hi2;
            lo2[0] = (half_t)__builtin_amdgcn_fmed3f(acc[j][i][0] + bv[j][0], lo, 65504.f); lo2[1] = (half_t)__builtin_amdgcn_fmed3f(acc[j][i][1] + bv[j][1], lo, 65504.f);
            hi2[0] = (half_t)__builtin_amdgcn_fmed3f(acc[j][i][2] + bv[j][2], lo, 65504.f); hi2[1] = (half_t)__builtin_amdgcn_fmed3f(acc[j][i][3] + bv[j][3], lo, 65504.f);
            pk[j][0] = __builtin_bit_cast(unsigned, lo2);
            pk[j][1] = __builtin_bit_cast(unsigned, hi2);
        }
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        const int z = (wr * NB + i) >> 2, y = 2 * ((wr * NB + i) & 3) + vyp;
        const int64_t orow = (((int64_t)b * p.D + z0 + z) * p.H + y0 + y) * p.W + x0 + vx;
        const int col = tn * TC + wc * 32 + (q & 1) * 16 + (q >> 1) * 8;
        if (p.resid != nullptr) {
            half8 ov = __builtin_bit_cast(half8, o);
            const half8 rs = *(const half8*)(p.resid + orow * p.Cout + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)ov[e] + (float)rs[e];
                if (p.relu) f = fmaxf(f, 0.f);
                ov[e] = to_half_sat(f);
            }
            o = __builtin_bit_cast(u32x4, ov);
        }
        *(u32x4*)(p.out + orow * p.Cout + col) = o;
    }
}

// one thread per 16-byte fragment piece: out[tile][stage][wc][ks][j][lane][8] = w[tile * tc + wc * 32 + j * 16 + (lane & 15)][k ..], k = tap * cin + kh * 64 + ks * 32 +
// (lane >> 4) * 8 for stage = tap * spt + kh (spt = stages per tap: cin / 64, or 1); the LAST stage = the cin2 weight columns of a second source behind the 27 taps
__global__ __launch_bounds__(256) void conv3d_pack_wfrag_kernel(const half_t* __restrict__ w, int kpad, int cin, int cout, int cin2, int nwc, half_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int spt = cin > 64 ? cin / 64 : 1, kss = (cin > 64 ? 64 : cin) / 32, nstg = 27 * spt + 1, tc = 32 * nwc;
    const int total = (cout / tc) * nstg * nwc * kss * 2 * 64;
    if (idx >= total) return;
    const int lane = idx & 63; int r = idx >> 6;
    const int j = r & 1; r >>= 1;
    const int ks = r % kss; r /= kss;
    const int wc = r % nwc; r /= nwc;
    const int g = r % nstg; const int tile = r / nstg;
    const int n = tile * tc + wc * 32 + j * 16 + (lane & 15);
    const int kk = ks * 32 + (lane >> 4) * 8;
    const bool extra = g == nstg - 1;
    const int k = extra ? 27 * cin + kk : (g / spt) * cin + (g % spt) * 64 + kk;
    const bool real = !extra || kk + 8 <= cin2;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    *(half8*)(out + (int64_t)idx * 8) = (real && k + 8 <= kpad) ? *(const half8*)(w + (int64_t)n * kpad + k) : zero8;
}

// split-K finish: sum the slabs in split order (deterministic), then the same epilogue as above.
// thread = (variant, row, 8-column chunk)
__global__ __launch_bounds__(256) void conv3d_finish_kernel(ConvParams p) {
    // grid = (chunks of a variant, variant): 32-bit index arithmetic (64-bit divisions by run-time values were most of this kernel)
    const unsigned cpr = (unsigned)p.Cout / 8;
    const unsigned rem = blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= (unsigned)p.M * cpr) return;
    const int cls = blockIdx.y;
    const int m = (int)(rem / cpr), col = (int)(rem - (unsigned)m * cpr) * 8;
    const ConvVariant& cv = p.var[cls];
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    // slabs added in split order (deterministic), eight splits' loads in flight together: the small-grid layers that
    // split K deepest (up to 64 ways) have the fewest threads here, so this loop is pure load latency
    const int64_t sstride = (int64_t)p.nvar * p.M * p.Cout;
    const float* sp = p.slabs + ((int64_t)cls * p.M + m) * p.Cout + col;
    constexpr int FL = 8;
    for (int s = 0; s < p.splits; s += FL) {
        float4 u[FL], v[FL];
#pragma unroll
        for (int j = 0; j < FL; ++j) {
            const int sj = s + j < p.splits ? s + j : s;
            const float4* q = (const float4*)(sp + (int64_t)sj * sstride);
            u[j] = q[0]; v[j] = q[1];
        }
#pragma unroll
        for (int j = 0; j < FL; ++j)
            if (s + j < p.splits) {
                a[0] += u[j].x; a[1] += u[j].y; a[2] += u[j].z; a[3] += u[j].w;
                a[4] += v[j].x; a[5] += v[j].y; a[6] += v[j].z; a[7] += v[j].w;
            }
    }
    const int ox = m % p.Wo; int t = m / p.Wo;
    const int oy = t % p.Ho; t /= p.Ho;
    const int oz = t % p.Do; const int b = t / p.Do;
    const int64_t orow = (((int64_t)b * p.OD + oz * p.os + cv.pz) * p.OH + oy * p.os + cv.py) * p.OW + ox * p.os + cv.px;
    half8 o;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (p.bias != nullptr) {
        const f32x4 b0 = *(const f32x4*)(p.bias + col), b1 = *(const f32x4*)(p.bias + col + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
    }
    if (p.resid != nullptr) {
        const half8 rs = *(const half8*)(p.resid + orow * p.Cout + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // same rounding points as the unsplit epilogue: fp16(acc + bias), then + residual in fp32
            float f = (float)to_half_sat(a[e] + bv[e]) + (float)rs[e];
            if (p.relu) f = fmaxf(f, 0.f);
            o[e] = to_half_sat(f);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = a[e] + bv[e];
            if (p.relu) f = fmaxf(f, 0.f);
            o[e] = to_half_sat(f);
        }
    }
    *(half8*)(p.out + orow * p.Cout + col) = o;
}

// first layer: x fp32 [B][D][H][W] (Cin = 1), k3 s1 p1 -> fp16 NDHWC [..][cout], ReLU.
// thread = (voxel, 8-channel chunk); w fp32 [cout][27], b fp32 [cout]
__global__ __launch_bounds__(256) void conv3d_first_kernel(const float* __restrict__ x, int B, int D, int H, int W,
                                                            int stride, const float* __restrict__ w,
                                                            const float* __restrict__ b, int cout,
                                                            half_t* __restrict__ out) {
    extern __shared__ float ws[];   // [cout][27] + [cout]
    for (int i = threadIdx.x; i < cout * 28; i += blockDim.x) ws[i] = i < cout * 27 ? w[i] : b[i - cout * 27];
    __syncthreads();
    const int chunks = cout / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;   // k3, pad 1
    const int64_t nvox = (int64_t)B * Do * Ho * Wo;
    if (idx >= nvox * chunks) return;
    const int64_t vox = idx / chunks;
    const int ch = (int)(idx - vox * chunks);
    int xx = (int)(vox % Wo); int64_t t = vox / Wo;
    int yy = (int)(t % Ho); t /= Ho;
    int zz = (int)(t % Do); const int bb = (int)(t / Do);
    xx *= stride; yy *= stride; zz *= stride;
    float tap[27];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iz = zz + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                const bool ok = (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                tap[(kz * 3 + ky) * 3 + kx] = ok ? x[(((int64_t)bb * D + iz) * H + iy) * W + ix] : 0.f;
            }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 8 + e;
        float a = ws[cout * 27 + c];
#pragma unroll
        for (int k = 0; k < 27; ++k) a = fmaf(ws[c * 27 + k], tap[k], a);
        o[e] = to_half_sat(fmaxf(a, 0.f));
    }
    *(half8*)(out + vox * cout + ch * 8) = o;
}

// first layer, cout = 32 fast path: thread = voxel, all 32 output channels.  Weight indices are wave-uniform, so
// they are scalar loads and the 864 FMAs per voxel take an SGPR operand (the generic kernel above reads one LDS
// word per FMA, which is what bounds it).
__global__ __launch_bounds__(256) void conv3d_first32_kernel(const float* __restrict__ x, int B, int D, int H, int W,
                                                              int stride, const float* __restrict__ w,
                                                              const float* __restrict__ b, half_t* __restrict__ out) {
    constexpr int COUT = 32;
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;   // k3, pad 1
    const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= (int64_t)B * Do * Ho * Wo) return;
    int xx = (int)(vox % Wo); int64_t t = vox / Wo;
    int yy = (int)(t % Ho); t /= Ho;
    int zz = (int)(t % Do); const int bb = (int)(t / Do);
    xx *= stride; yy *= stride; zz *= stride;
    float tap[27];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iz = zz + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                const bool ok = (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int cz = min(max(iz, 0), D - 1), cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
                const float v = x[(((int64_t)bb * D + cz) * H + cy) * W + cx];
                tap[(kz * 3 + ky) * 3 + kx] = ok ? v : 0.f;
            }
    half_t* o = out + vox * COUT;
#pragma unroll
    for (int c8 = 0; c8 < COUT / 8; ++c8) {
        half8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c8 * 8 + e;
            float a = b[c];
#pragma unroll
            for (int k = 0; k < 27; ++k) a = fmaf(w[c * 27 + k], tap[k], a);
            r[e] = to_half_sat(fmaxf(a, 0.f));
        }
        *(half8*)(o + c8 * 8) = r;
    }
}

// encoder.0 on the matrix cores (stride 1, cout = 32, D % 4 == H % 4 == W % 8 == 0).  The VALU form above spends 864 fp32 FMAs
// per voxel (27 taps x 32 channels): 57 us at B = 32, five times what writing the 67 MB of output takes.  Here the layer is
// the product D[channel][voxel] = W[channel][tap] . im2col[tap][voxel] with K = 27 taps padded to one 32-deep MFMA step:
// weights (rounded to fp16 like every other layer's) are the A operand, held in registers; a workgroup keeps the fp16 image of
// its 6 x 6 x 10 input halo in LDS (720 B) and a lane gathers the eight taps of its k chunk for its voxel from it; the
// accumulator of a lane is 4 consecutive channels of one voxel = an 8-byte piece of the NDHWC output row.
// TZ x TY x 8 output voxels per tile (4 x 4 x 8, or 8 x 8 x 8 where the grid divides: the kernel is bound by instruction ISSUE -- 470 instructions per wave and
// 4 x 4 x 8 tile, of which 120 stage the halo and 60 decode the tile index; with 512 voxels per tile they are spread over four times the MFMA groups: 31 -> ~20 us)
template <int TZ, int TY>
__global__ __launch_bounds__(256) void conv3d_first32_mfma_kernel(const float* __restrict__ x, int B, int D, int H, int W,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias, half_t* __restrict__ out,
                                                                   int ntz, int nty, int ntx, int ntiles) {
    constexpr int HTZ = TZ, HTY = TY, HHY = HTY + 2, HROWS = (HTZ + 2) * HHY * HHX, GROUPS = TZ * TY * 8 / 16 / 4, YB = TY == 8 ? 3 : 2;   // (shadow the file's 128-row geometry)
    __shared__ half_t halo[2][HROWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    // weights and tap offsets once per workgroup (it walks several tiles)
    half8 wa[2];
    int off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int tap = 8 * q + j;                         // taps 27..31 are K padding: zero weights, any finite input
        const bool real = tap < 27;
        const int tc = real ? tap : 0;
        wa[0][j] = real ? (half_t)w[r * 27 + tc] : (half_t)0.f;
        wa[1][j] = real ? (half_t)w[(16 + r) * 27 + tc] : (half_t)0.f;
        off[j] = ((tc / 9) * HHY + (tc / 3) % 3) * HHX + tc % 3;
    }
    const f32x4 b0 = *(const f32x4*)(bias + 4 * q), b1 = *(const f32x4*)(bias + 16 + 4 * q);
    int buf = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        int t = tile;
        const int tx = t % ntx; t /= ntx;
        const int ty = t % nty; t /= nty;
        const int tz = t % ntz; const int b = t / ntz;
        const int z0 = tz * HTZ, y0 = ty * HTY, x0 = tx * HTX;
        constexpr int SIT = (HROWS + 255) / 256;
        float hv[SIT];
#pragma unroll
        for (int it = 0; it < SIT; ++it) {                 // every load of the halo in flight before the first LDS store
            const int i = it * 256 + tid;
            const int hx = i % HHX; const int r2 = i / HHX;
            const int hy = r2 % HHY, hz = r2 / HHY;
            const int iz = z0 - 1 + hz, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = i < HROWS && (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int cz = min(max(iz, 0), D - 1), cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const float v = x[(((int64_t)b * D + cz) * H + cy) * W + cx];
            hv[it] = ok ? v : 0.f;
        }
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int i = it * 256 + tid;
            if (i < HROWS) halo[buf][i] = (half_t)hv[it];
        }
        __syncthreads();                                   // two halo buffers: one barrier per tile is enough
#pragma unroll
        for (int i = 0; i < GROUPS; ++i) {
            const int m = (wave * GROUPS + i) * 16 + r;    // this lane's voxel: the MFMA column
            const int vx = m & 7, vy = (m >> 3) & (TY - 1), vz = m >> (3 + YB);
            const int base = (vz * HHY + vy) * HHX + vx;
            half8 col;
#pragma unroll
            for (int j = 0; j < 8; ++j) col[j] = halo[buf][base + off[j]];
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0], col, zero, 0, 0, 0);
            const f32x4 a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1], col, zero, 0, 0, 0);
            // lane (voxel, q) holds channels 4q..4q+3 of each 16-channel block; the two blocks exchange halves between the
            // 16-lane groups (v_permlane16_swap, as in the GEMM epilogue): group q then owns 8 consecutive channels
            unsigned pk[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                half2_ h0, h1;
                h0[0] = (half_t)__builtin_amdgcn_fmed3f(a0[2 * e] + b0[2 * e], 0.f, 65504.f);
                h0[1] = (half_t)__builtin_amdgcn_fmed3f(a0[2 * e + 1] + b0[2 * e + 1], 0.f, 65504.f);
                h1[0] = (half_t)__builtin_amdgcn_fmed3f(a1[2 * e] + b1[2 * e], 0.f, 65504.f);
                h1[1] = (half_t)__builtin_amdgcn_fmed3f(a1[2 * e + 1] + b1[2 * e + 1], 0.f, 65504.f);
                pk[0][e] = __builtin_bit_cast(unsigned, h0);
                pk[1][e] = __builtin_bit_cast(unsigned, h1);
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            half_t* orow = out + ((((int64_t)b * D + z0 + vz) * H + y0 + vy) * W + x0 + vx) * 32;
            *(u32x4*)(orow + (q & 1) * 16 + (q >> 1) * 8) = o;
        }
    }
}

// last layer: fp16 NDHWC [..][CIN] -> fp32 [B][D][H][W], k3 s1 p1, Cout = 1, sigmoid.
// w fp32 [27][CIN], one thread per output voxel.
template <int CIN>
__global__ __launch_bounds__(256) void conv3d_last_kernel(const half_t* __restrict__ in, int B, int D, int H, int W,
                                                           const float* __restrict__ w, float bias,
                                                           float* __restrict__ out) {
    __shared__ float ws[27 * CIN];
    for (int i = threadIdx.x; i < 27 * CIN; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= (int64_t)B * D * H * W) return;
    const int xx = (int)(vox % W); int64_t t = vox / W;
    const int yy = (int)(t % H); t /= H;
    const int zz = (int)(t % D); const int bb = (int)(t / D);
    float a = bias;
    for (int kz = 0; kz < 3; ++kz)
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int iz = zz + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                if ((unsigned)iz >= (unsigned)D || (unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const half8* src = (const half8*)(in + ((((int64_t)bb * D + iz) * H + iy) * W + ix) * CIN);
                const float* wk = ws + ((kz * 3 + ky) * 3 + kx) * CIN;
#pragma unroll
                for (int c8 = 0; c8 < CIN / 8; ++c8) {
                    const half8 v = src[c8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a = fmaf(wk[c8 * 8 + e], (float)v[e], a);
                }
            }
    out[vox] = 1.f / (1.f + expf(-a));
}

// last layer with the halo in LDS: a workgroup owns 4 x 4 x 8 output voxels; thread = (voxel, half of the 27
// taps); the direct kernel above re-reads every input voxel 27 times from L2 (1.8 GB at B = 32).
// Requires D % 4 == H % 4 == W % 8 == 0.  Weights fp32 [27][32] are wave-uniform -> scalar loads.
__global__ __launch_bounds__(256) void conv3d_last_halo_kernel(const half_t* __restrict__ in, int B, int D, int H, int W,
                                                               const float* __restrict__ w, float bias,
                                                               float* __restrict__ out, int ntz, int nty, int ntx) {
    constexpr int CIN = 32, RB = CIN * 2, P = RB + 16, CPR = RB / 16;
    constexpr int HIT = (HROWS * CPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) char smem[HROWS * P];
    __shared__ float part[128];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz; const int b = t / ntz;
    const int z0 = tz * HTZ, y0 = ty * HTY, x0 = tx * HTX;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int c = it * 256 + tid;
        const int row = c / CPR, ch = c - row * CPR;
        const int hx = row % HHX; const int r2 = row / HHX;
        const int hy = r2 % HHY, hz = r2 / HHY;
        const int iz = z0 - 1 + hz, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < HROWS) {
            if (ok) v = *(const half8*)(in + ((((int64_t)b * D + iz) * H + iy) * W + ix) * CIN + ch * 8);
            *(half8*)(smem + row * P + ch * 16) = v;
        }
    }
    __syncthreads();
    const int vox = tid & 127, half = __builtin_amdgcn_readfirstlane(tid >> 7);   // waves 0,1: taps 0-13; 2,3: 14-26
    const int x = vox & 7, y = (vox >> 3) & 3, z = vox >> 5;
    const char* base = smem + ((z * HHY + y) * HHX + x) * P;
    float a = 0.f;
    const int t0 = half * 14, t1 = half ? 27 : 14;
    for (int tap = t0; tap < t1; ++tap) {
        const int toff = ((tap / 9) * HHY + (tap / 3) % 3) * HHX + tap % 3;
        const float* wk = w + tap * CIN;                          // uniform address: s_load
#pragma unroll
        for (int c8 = 0; c8 < CIN / 8; ++c8) {
            const half8 v = *(const half8*)(base + toff * P + c8 * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(wk[c8 * 8 + e], (float)v[e], a);
        }
    }
    if (half) part[vox] = a;
    __syncthreads();
    if (!half) {
        a += part[vox] + bias;
        out[(((int64_t)b * D + z0 + z) * H + y0 + y) * W + x0 + x] = 1.f / (1.f + expf(-a));
    }
}

// The same layer with an 8 x 8 x 8 output block per workgroup (512 threads = 512 voxels, all 27 taps per thread).  The 4 x 4 x 8 form above
// fetches every input voxel 2.8 times (6 x 6 x 10 halo per 128 outputs: 188 MB of L2 -> LDS traffic at B = 32) and that, not its arithmetic,
// is what bounds it (a packed-fp16 dot-product form ran in the same 61 us: profiles/r04_j); here the halo is 10 x 10 x 10 per 512 outputs = 1.95 x.
template <int TZ>
__global__ __launch_bounds__(64 * TZ) void conv3d_last_halo8_kernel(const half_t* __restrict__ in, int B, int D, int H, int W,
                                                                    const float* __restrict__ w, float bias,
                                                                    float* __restrict__ out, int ntz, int nty, int ntx) {
    // rows padded to 80 bytes (conflict-free 16-byte reads of x-adjacent voxels).  Unpadded 64-byte rows with an XOR swizzle (64 KB per workgroup: two
    // per CU) measured the same (52.9 v. 51.8 us): the layer is VALU-bound now (864 v_fma_mix_f32 per voxel = ~48 us at B = 32), profiles/r04_j
    constexpr int CIN = 32, RB = CIN * 2, P = RB + 16, CPR = RB / 16, T = 8, HH = T + 2, ROWS = (TZ + 2) * HH * HH, NTH = 64 * TZ;
    constexpr int HIT = (ROWS * CPR + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) char smem[ROWS * P];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz; const int b = t / ntz;
    const int z0 = tz * TZ, y0 = ty * T, x0 = tx * T;
    half8 hv[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {                           // every load of the halo in flight before the first LDS store
        const int c = it * NTH + tid;
        const int row = c / CPR, ch = c - row * CPR;
        const int hx = row % HH; const int r2 = row / HH;
        const int hy = r2 % HH, hz = r2 / HH;
        const int iz = z0 - 1 + hz, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = row < ROWS && (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        hv[it] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) hv[it] = *(const half8*)(in + ((((int64_t)b * D + iz) * H + iy) * W + ix) * CIN + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int c = it * NTH + tid;
        const int row = c / CPR, ch = c - row * CPR;
        if (row < ROWS) *(half8*)(smem + row * P + ch * 16) = hv[it];
    }
    __syncthreads();
    const int x = tid & 7, y = (tid >> 3) & 7, z = tid >> 6;
    const int r0 = (z * HH + y) * HH + x;
    float a = bias;
#pragma unroll 3
    for (int tap = 0; tap < 27; ++tap) {
        const int row = r0 + ((tap / 9) * HH + (tap / 3) % 3) * HH + tap % 3;
        const float* wk = w + tap * CIN;                          // uniform address: s_load
#pragma unroll
        for (int c8 = 0; c8 < CIN / 8; ++c8) {
            const half8 v = *(const half8*)(smem + row * P + c8 * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(wk[c8 * 8 + e], (float)v[e], a);
        }
    }
    out[(((int64_t)b * D + z0 + z) * H + y0 + y) * W + x0 + x] = 1.f / (1.f + expf(-a));
}

// fp32 weights [27][32] -> the MFMA A operand of every tap for conv3d_last_mfma8_kernel: [27][64 lanes][8 halfs]
__global__ __launch_bounds__(64) void conv3d_last_pack_kernel(const float* __restrict__ w, half_t* __restrict__ wfrag) {
    const int tap = blockIdx.x, lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float wf = w[tap * 32 + kq * 8 + e];
        const half_t hi = (half_t)wf;
        const half_t lo = (half_t)(wf - (float)hi);
        const half_t lo2 = (half_t)((wf - (float)hi) - (float)lo);
        v[e] = m == 0 ? hi : (m == 1 ? lo : (m == 2 ? lo2 : (half_t)0.f));
    }
    *(half8*)(wfrag + (tap * 64 + lane) * 8) = v;
}

// The same 8 x 8 x 8 block with the 864 multiply-adds per voxel on the MATRIX pipe (round 5).  The layer has ONE output channel, so as a product it is
// D[3][voxel] = Wt[3][K = 27 x 32] . X[K][voxel] with the three rows = the fp16 head of the fp32 weights and two successive fp16 rounding residuals (hi + lo + lo2: the
// fp32 weights to their last bit, fp32 accumulation; the other 13 rows of the 16 x 16 x 32 MFMA are zeros -- wasted, and still 4 x faster than 864 v_fma_mix per voxel:
// 108 MFMAs of 16 cycles per wave and 64 voxels against ~3500 VALU cycles).  A k step is one tap: the B operand of voxel n is its 32 input channels at the tap's
// halo row (the same 16-byte LDS reads as the VALU form); the A operand of tap t is 1 KB of a fragment-order copy of the weights (pcd_conv3d_last_pack, made once).
// Lanes 0-15 hold rows 0-2 (the three partial sums) of their voxel: out = sigmoid(hi + (lo + lo2) + bias).
__global__ __launch_bounds__(512, 4) void conv3d_last_mfma8_kernel(const half_t* __restrict__ in, int B, int D, int H, int W, const half_t* __restrict__ wfrag,
                                                                   float bias, float* __restrict__ out, int ntz, int nty, int ntx) {
    constexpr int CIN = 32, RB = CIN * 2, P = RB + 16, CPR = RB / 16, T = 8, TZ = 8, HH = T + 2, ROWS = (TZ + 2) * HH * HH, NTH = 64 * TZ;
    constexpr int HIT = (ROWS * CPR + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) char smem[ROWS * P];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    int t = blockIdx.x;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz; const int b = t / ntz;
    const int z0 = tz * TZ, y0 = ty * T, x0 = tx * T;
    {
        half8 hv[HIT];
#pragma unroll
        for (int it = 0; it < HIT; ++it) {                       // every load of the halo in flight before the first LDS store
            const int c = it * NTH + tid;
            const int row = c / CPR, ch = c - row * CPR;
            const int hx = row % HH; const int r2 = row / HH;
            const int hy = r2 % HH, hz = r2 / HH;
            const int iz = z0 - 1 + hz, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = row < ROWS && (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            hv[it] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) hv[it] = *(const half8*)(in + ((((int64_t)b * D + iz) * H + iy) * W + ix) * CIN + ch * 8);
        }
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int c = it * NTH + tid;
            const int row = c / CPR, ch = c - row * CPR;
            if (row < ROWS) *(half8*)(smem + row * P + ch * 16) = hv[it];
        }
    }
    // A operand of tap t: wfrag[t][lane][8] (conv3d_last_pack_kernel: lane (m = lane & 15, k = 8 (lane >> 4) .. + 7), rows 0-2 = hi / lo / lo2, rows 3-15 zero),
    // one coalesced 1-KB load per tap and wave from 27 KB that every workgroup reads (L1 / L2 resident)
    const int m = lane & 15, kq = lane >> 4;
    const half_t* wl = wfrag + lane * 8;
    __syncthreads();
    const int z = tid >> 6;                                       // the wave's z slice: 64 voxels = four groups of 16 (two x rows of 8 each)
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
    for (int tap = 0; tap < 27; ++tap) {
        const int toff = ((tap / 9) * HH + (tap / 3) % 3) * HH + tap % 3;
        const half8 wa = *(const half8*)(wl + tap * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int v = g * 16 + m, x = v & 7, y = v >> 3;
            const half8 xb = *(const half8*)(smem + ((z * HH + y) * HH + x + toff) * P + kq * 16);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb, acc[g], 0, 0, 0);
        }
    }
    if (kq == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int v = g * 16 + m, x = v & 7, y = v >> 3;
            const float a = (acc[g][0] + (acc[g][1] + acc[g][2])) + bias;
            out[(((int64_t)b * D + z0 + z) * H + y0 + y) * W + x0 + x] = 1.f / (1.f + expf(-a));
        }
    }
}

// The same layer as PER-TAP PARTIAL PRODUCTS (round 5, default).  The form above reads one 1-KB B fragment from LDS per MFMA: 108 KB per wave and 64 voxels, the LDS
// port is busy 3.2 us per workgroup of a 2048-workgroup grid (26 us) and the layer took 51-56 us for 0.9 GFLOP.  Here a wave owns one 8 x 8 output slice and turns the
// product inside out: for each of the three input slices above it (dz = -1, 0, +1) it forms, for every voxel u of that slice's 10 x 10 halo,
//       p[u][t] = sum_c w[dz][t][c] x[u][c]        for the nine taps t = 3 dy + dx of that dz
// as D[tap][voxel] = W[tap][K = 32 c] . X[c][voxel] on v_mfma_f32_16x16x32_f16 -- the A operand holds nine tap rows (the fp32 weights as hi + lo + lo2 fp16 parts,
// three MFMAs into one accumulator), the B operand 16 voxels x 32 channels straight from global memory (64 contiguous bytes per voxel: no input staging) -- writes
// the 16 x 9 partials to a WAVE-PRIVATE 5-KB LDS tile and gathers out[y][x] += p[(y + dy, x + dx)][3 dy + dx]: nine 4-byte reads per dz and lane, a fixed
// summation order.  21 B-fragment loads (all in flight), 63 MFMAs, 27 + 21 LDS operations per wave; no workgroup barrier (LDS operations of one wave execute in order).
// Each input slice is read by the three waves around it: L1 / L2 absorb that.  d a multiple of 4, h and w of 8.
template <int TZ, int ABL = 0>  // TZ: output slices = waves per workgroup (4: three workgroups per CU at this kernel's 131 registers); ABL (tools/bench_last_layer.py,
                                // outputs wrong): 1 = loads only (their sum instead of products and gathers), 2 = no loads
__global__ __launch_bounds__(64 * TZ, 4) void conv3d_last_taps_kernel(const half_t* __restrict__ in, int B, int D, int H, int W, const half_t* __restrict__ wfrag, float bias,
                                                               float* __restrict__ out, int ntz, int nty, int ntx) {
    constexpr int CIN = 32, HH = 10, NV = HH * HH, NG = (NV + 15) / 16, PITCH = 12;
    __shared__ __attribute__((aligned(16))) float psm[TZ][NG * 16][PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz; const int b = t / ntz;
    const int z0 = tz * TZ, y0 = ty * 8, x0 = tx * 8;
    const int n = lane & 15, q = lane >> 4;
    // A operands: row m = n is tap 3 dy + dx of slice dz (rows 9 .. 15 zero), k = 8 q .. 8 q + 7; three fp16 parts of the fp32 weight
    // (from pcd_conv3d_last_pack's copy [27 taps][64 lanes][8]: there lane part + 16 q of tap t holds part `part` of w[t][8 q .. 8 q + 7]); one slice's set at a time
    half8 wa[3];
    auto load_w = [&](int dz) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            wa[part] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (n < 9) wa[part] = *(const half8*)(wfrag + ((dz * 9 + n) * 64 + part + 16 * q) * 8);
        }
    };
    load_w(0);
    float (*pw)[PITCH] = psm[wave];
    const int oy = lane >> 3, ox = lane & 7;
    float o = bias;
    // the B fragments of the first two input slices are requested before the first is used; the third slice's take the first slice's registers behind its MFMAs
    // (56 + 12 weight registers: four waves per SIMD).  A lane's in-plane offsets and bounds are the same for the three slices: formed once; the loads are raw
    // buffer loads over the slice (a lane outside the grid carries an offset past its end and receives zeros: no branch, no per-load address arithmetic).
    half8 xb[3][NG];
    unsigned voff[NG];                                               // byte offset inside a slice, or past the end of every slice: the buffer load returns zeros there
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int v = g * 16 + n, hy = v / HH, hx = v - hy * HH;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = !(ABL & 2) && v < NV && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        voff[g] = ok ? (unsigned)(((iy * W + ix) * CIN + 8 * q) * 2) : 0xffffffffu;
    }
    auto request = [&](int dz) __attribute__((always_inline)) {
        const int iz = z0 + wave - 1 + dz;
        const bool zok = (unsigned)iz < (unsigned)D;                 // wave-uniform: a slice outside the grid is a buffer of zero bytes
        const half_t* slice = in + ((int64_t)b * D + (zok ? iz : 0)) * H * W * CIN;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slice, 0, zok ? H * W * CIN * 2 : 0, 0x00020000);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[g], 0, 0);
            xb[dz][g] = __builtin_bit_cast(half8, r);
        }
    };
    request(0);
    request(1);
    if constexpr ((ABL & 1) != 0) request(2);
    if constexpr ((ABL & 1) != 0) {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int g = 0; g < NG; ++g) o += (float)xb[dz][g][0] + (float)xb[dz][g][7];
        out[(((int64_t)b * D + z0 + wave) * H + y0 + oy) * W + x0 + ox] = o;
        return;
    }
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
        f32x4 accs[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[2], xb[dz][g], acc, 0, 0, 0);      // smallest part first
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1], xb[dz][g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0], xb[dz][g], acc, 0, 0, 0);
            accs[g] = acc;
        }
        // lane (voxel n, q) holds taps 4 q .. 4 q + 3 of voxel g * 16 + n (taps 9 .. 11 are zero rows of the A operand; taps 12 .. 15 have no place in the tile)
        if (q < 3) {
#pragma unroll
            for (int g = 0; g < NG; ++g) *(f32x4*)&pw[g * 16 + n][4 * q] = accs[g];
        }
        if (dz < 2) {
            __builtin_amdgcn_sched_barrier(0);                       // (not earlier: the compiler would hoist these loads to the top and need 50 more registers)
            load_w(dz + 1);
            if (dz == 0) request(2);
            __builtin_amdgcn_sched_barrier(0);
        }
        // Lanes exchange data through the tile: to the language that is communication between threads and needs synchronisation, or the compiler may (and did)
        // keep a lane's reads in front of stores the lane itself does not execute.  Within one wave the hardware needs nothing (its LDS operations execute in
        // order): wavefront-scope fences + a wave barrier emit no instruction and pin the order.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) o += pw[(oy + dy) * HH + ox + dx][3 * dy + dx];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next slice's stores stay behind these reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    out[(((int64_t)b * D + z0 + wave) * H + y0 + oy) * W + x0 + ox] = 1.f / (1.f + expf(-o));
}

// The same per-tap partial products with every input slice read ONCE: a wave owns TZ consecutive output slices of one 8 x 8 tile and walks the TZ + 2 input
// slices around them; for a slice it forms ALL 27 taps' partials (two MFMA row groups: taps 0-15 and 16-26, three weight parts each: six MFMAs per 16 voxels),
// writes them to its LDS tile ([112 voxels][28 taps]) and every output slice the slice touches (z = s + 1 - dz) gathers its nine.  The kernel above issues ~650
// instructions per output slice (it is bound by instruction issue); here the loads, their offsets and the prologue are shared by TZ slices: ~300 per slice at TZ = 4
// (22.7 us against 32; TZ = 2: 26.6, TZ = 8: 24.4 -- 2048 waves do not fill the chip).
// The next slice's B fragments are requested before the current slice's MFMAs.  Summation order per output: input slice ascending, then (dy, dx): deterministic.
template <int TZ>
__global__ __launch_bounds__(256, 2) void conv3d_last_roll_kernel(const half_t* __restrict__ in, int B, int D, int H, int W, const half_t* __restrict__ wfrag, float bias,
                                                                  float* __restrict__ out, int ntz, int nty, int ntx, int nwork) {
    constexpr int CIN = 32, HH = 10, NV = HH * HH, NG = (NV + 15) / 16, PITCH = 28;
    __shared__ __attribute__((aligned(16))) float psm[4][NG * 16][PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x * 4 + wave;                                   // one (sample, z chunk, y tile, x tile) per wave; no workgroup-level synchronisation below
    if (t >= nwork) return;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz; const int b = t / ntz;
    const int z0 = tz * TZ, y0 = ty * 8, x0 = tx * 8;
    const int n = lane & 15, q = lane >> 4;
    // A operands: row group rg, row m = n is tap 16 rg + n (taps 27 .. 31: zero rows), k = 8 q ..; from pcd_conv3d_last_pack's copy (lane part + 16 q of tap t)
    half8 wa[2][3];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            wa[rg][part] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (16 * rg + n < 27) wa[rg][part] = *(const half8*)(wfrag + ((16 * rg + n) * 64 + part + 16 * q) * 8);
        }
    unsigned voff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int v = g * 16 + n, hy = v / HH, hx = v - hy * HH;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = v < NV && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        voff[g] = ok ? (unsigned)(((iy * W + ix) * CIN + 8 * q) * 2) : 0xffffffffu;
    }
    half8 xb[2][NG];
    auto request = [&](int s, half8 (&dst)[NG]) __attribute__((always_inline)) {      // input slice z0 - 1 + s
        const int iz = z0 - 1 + s;
        const bool zok = (unsigned)iz < (unsigned)D;                 // wave-uniform: a slice outside the grid is a buffer of zero bytes
        const half_t* slice = in + ((int64_t)b * D + (zok ? iz : 0)) * H * W * CIN;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slice, 0, zok ? H * W * CIN * 2 : 0, 0x00020000);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            dst[g] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[g], 0, 0));
        }
    };
    float (*pw)[PITCH] = psm[wave];
    const int oy = lane >> 3, ox = lane & 7;
    float o[TZ];
#pragma unroll
    for (int z = 0; z < TZ; ++z) o[z] = bias;
    request(0, xb[0]);
#pragma unroll
    for (int s = 0; s < TZ + 2; ++s) {
        if (s + 1 < TZ + 2) request(s + 1, xb[(s + 1) & 1]);
        // lane (voxel n, q) holds taps 16 rg + 4 q .. + 3 of voxel g * 16 + n; taps 28 .. 31 (rg 1, q 3) have no place in the tile
        f32x4 acc1[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0][2], xb[s & 1][g], acc, 0, 0, 0);      // smallest part first
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0][1], xb[s & 1][g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0][0], xb[s & 1][g], acc, 0, 0, 0);
            *(f32x4*)&pw[g * 16 + n][4 * q] = acc;
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1][2], xb[s & 1][g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1][1], xb[s & 1][g], acc, 0, 0, 0);
            acc1[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1][0], xb[s & 1][g], acc, 0, 0, 0);
        }
        if (q < 3) {
#pragma unroll
            for (int g = 0; g < NG; ++g) *(f32x4*)&pw[g * 16 + n][16 + 4 * q] = acc1[g];
        }
        // lanes exchange data through the tile: wavefront-scope fences + a wave barrier pin the order for the compiler (conv3d_last_taps_kernel)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            const int z = s - dz;                                    // output slice z0 + z reads input slice z0 - 1 + s with its taps of dz
            if (z >= 0 && z < TZ) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) o[z] += pw[(oy + dy) * HH + ox + dx][9 * dz + 3 * dy + dx];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next slice's stores stay behind these reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int z = 0; z < TZ; ++z) out[(((int64_t)b * D + z0 + z) * H + y0 + oy) * W + x0 + ox] = 1.f / (1.f + expf(-o[z]));
}

// VAE3D's last layer (networks.py:2018-2019): ConvTranspose3d(CIN, 1, k3, s2, p1, output_padding 1) + Sigmoid.
// o = 2 i - 1 + k per dimension: even o takes (k=1, i=o/2); odd o takes (k=0, i=(o+1)/2) and (k=2, i=(o-1)/2).
// in fp16 NDHWC [B][D][H][W][CIN]; w fp32 [27][CIN] (tap-major); out fp32 [B][2D][2H][2W].
template <int CIN>
__global__ __launch_bounds__(256) void convT3d_last_kernel(const half_t* __restrict__ in, int B, int D, int H, int W,
                                                            const float* __restrict__ w, float bias,
                                                            float* __restrict__ out) {
    __shared__ float ws[27 * CIN];
    for (int i = threadIdx.x; i < 27 * CIN; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
    const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= (int64_t)B * OD * OH * OW) return;
    const int ox = (int)(vox % OW); int64_t t = vox / OW;
    const int oy = (int)(t % OH); t /= OH;
    const int oz = (int)(t % OD); const int bb = (int)(t / OD);
    float a = bias;
    for (int kz = (oz & 1) ? 0 : 1; kz < 3; kz += 2) {
        const int iz = (oz + 1 - kz) >> 1;
        if ((unsigned)iz >= (unsigned)D) continue;
        for (int ky = (oy & 1) ? 0 : 1; ky < 3; ky += 2) {
            const int iy = (oy + 1 - ky) >> 1;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = (ox & 1) ? 0 : 1; kx < 3; kx += 2) {
                const int ix = (ox + 1 - kx) >> 1;
                if ((unsigned)ix >= (unsigned)W) continue;
                const half8* src = (const half8*)(in + ((((int64_t)bb * D + iz) * H + iy) * W + ix) * CIN);
                const float* wk = ws + ((kz * 3 + ky) * 3 + kx) * CIN;
#pragma unroll
                for (int c8 = 0; c8 < CIN / 8; ++c8) {
                    const half8 v = src[c8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a = fmaf(wk[c8 * 8 + e], (float)v[e], a);
                }
            }
        }
    }
    out[vox] = 1.f / (1.f + expf(-a));
}

}  // namespace pcd

using namespace pcd;

// split-K factor for a launch of `blocks` workgroups over `nk` K tiles: aim for ~2 workgroups per CU
static int g_wreg_abl = 0;    // timing ablations of conv3d_halo_wreg_kernel<64, 2> (pcd_conv3d_config + 32768 x bits; outputs wrong)
static int g_igemm_abl = 0;   // timing ablations of the 128 x 128 implicit GEMM (pcd_conv3d_config + 1024 / + 2048; outputs wrong)
static int g_split_target = 512;      // split-K aims at this many workgroups (tuning hook: pcd_conv3d_config + 64: 384, + 32: 768, + 96: 1024; VAE3DLarge encode at B = 32:
                                      // 834 / 913 / 888 us against 817 at 512: tools/bench_vae.py with PCD_CONV3D_CONFIG)
static int conv_splits(int64_t blocks, int nk) {
    if (blocks >= g_split_target || nk < 8) return 1;
    int s = (int)ceil_div((int64_t)g_split_target, blocks);
    s = s < nk / 4 ? s : nk / 4;
    s = s < 64 ? s : 64;
    return s < 1 ? 1 : s;
}

static int conv_check(const pcd_conv3d_desc_t* d) {
    PCD_CHECK_ARG(d->in && d->w && d->out && d->taps && d->zero_page);
    PCD_CHECK_ARG(d->batch > 0 && d->in_d > 0 && d->in_h > 0 && d->in_w > 0);
    PCD_CHECK_ARG(d->cin >= 8 && (d->cin & (d->cin - 1)) == 0);
    PCD_CHECK_ARG(d->cout > 0 && d->cout % 8 == 0);
    PCD_CHECK_ARG(d->ntaps > 0 && d->ntaps <= MAX_TAPS && d->kpad % CBK == 0 && d->kpad >= d->ntaps * d->cin);
    PCD_CHECK_ARG(d->kpad / d->cin <= TAP_SLOTS);
    // the gather tests all six grid bounds in packed byte fields (conv3d_igemm_kernel): coordinates below 64
    PCD_CHECK_ARG(d->in_d <= 64 && d->in_h <= 64 && d->in_w <= 64);
    PCD_CHECK_ARG((d->rows_d - 1) * d->stride < 64 && (d->rows_h - 1) * d->stride < 64 && (d->rows_w - 1) * d->stride < 64);
    PCD_CHECK_ARG((int64_t)d->batch * d->in_d * d->in_h * d->in_w * d->cin <= 0x7fffffff);
    PCD_CHECK_ARG(d->rows_d > 0 && d->rows_h > 0 && d->rows_w > 0 && d->stride > 0);
    PCD_CHECK_ARG(d->out_scale > 0 && d->out_d > 0 && d->out_h > 0 && d->out_w > 0);
    PCD_CHECK_ARG((int64_t)d->batch * d->rows_d * d->rows_h * d->rows_w <= 0x7fffffff);
    if (d->in2 != nullptr) {
        // second source: the row grid is its grid, its K columns start on a K-tile boundary behind the taps
        PCD_CHECK_ARG(d->cin2 >= 32 && (d->cin2 & (d->cin2 - 1)) == 0 && d->resid == nullptr);
        PCD_CHECK_ARG(d->stride == 1 && d->out_scale == 1 && d->rows_d == d->in_d && d->rows_h == d->in_h && d->rows_w == d->in_w);
        PCD_CHECK_ARG((d->ntaps * d->cin) % CBK == 0 && d->kpad >= d->ntaps * d->cin + d->cin2);
        PCD_CHECK_ARG((int64_t)d->batch * d->in_d * d->in_h * d->in_w * d->cin2 <= 0x7fffffff);
    }
    return PCD_OK;
}

static void conv_shape(const pcd_conv3d_desc_t* d, int n, int64_t* m, int* tiles_n, int* splits) {
    *m = (int64_t)d->batch * d->rows_d * d->rows_h * d->rows_w;
    *tiles_n = (int)ceil_div(d->cout, d->cout <= 64 ? 64 : 128);
    *splits = conv_splits(ceil_div(*m, 128) * *tiles_n * n, d->kpad / CBK);
}

extern "C" size_t pcd_conv3d_workspace_bytes(const pcd_conv3d_desc_t* d, int n) {
    if (d == nullptr || n < 1 || n > MAX_VARIANTS || conv_check(d) != PCD_OK) return 0;
    int64_t m; int tiles_n, splits;
    conv_shape(d, n, &m, &tiles_n, &splits);
    return splits > 1 ? (size_t)splits * n * m * d->cout * sizeof(float) : 0;
}

extern "C" int pcd_conv3d_f16_multi(const pcd_conv3d_desc_t* descs, int n, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    PCD_CHECK_ARG(descs != nullptr && n >= 1 && n <= MAX_VARIANTS);
    const pcd_conv3d_desc_t* d = descs;
    for (int i = 0; i < n; ++i) {
        const pcd_conv3d_desc_t* e = descs + i;
        const int rc = conv_check(e);
        if (rc != PCD_OK) return rc;
        // one launch: everything except the tap table, the weights and the output parity is shared
        PCD_CHECK_ARG(e->in == d->in && e->batch == d->batch && e->in_d == d->in_d && e->in_h == d->in_h &&
                      e->in_w == d->in_w && e->cin == d->cin && e->rows_d == d->rows_d && e->rows_h == d->rows_h &&
                      e->rows_w == d->rows_w && e->stride == d->stride && e->ntaps == d->ntaps && e->kpad == d->kpad &&
                      e->bias == d->bias && e->resid == d->resid && e->relu == d->relu && e->out == d->out &&
                      e->cout == d->cout && e->out_d == d->out_d && e->out_h == d->out_h && e->out_w == d->out_w &&
                      e->out_scale == d->out_scale && e->in2 == d->in2 && e->cin2 == d->cin2);
    }
    // the implicit GEMM reads a second source in whole K tiles, with no padding behind it
    PCD_CHECK_ARG(d->in2 == nullptr || (d->cin2 % CBK == 0 && d->kpad == d->ntaps * d->cin + d->cin2));
    ConvParams p{};
    p.in = (const half_t*)d->in; p.D = d->in_d; p.H = d->in_h; p.W = d->in_w; p.Cin = d->cin;
    p.cin_shift = __builtin_ctz((unsigned)d->cin);
    p.Do = d->rows_d; p.Ho = d->rows_h; p.Wo = d->rows_w; p.stride = d->stride;
    p.ntaps = d->ntaps; p.kpad = d->kpad;
    p.bias = d->bias; p.resid = (const half_t*)d->resid;
    p.out = (half_t*)d->out; p.Cout = d->cout;
    p.OD = d->out_d; p.OH = d->out_h; p.OW = d->out_w; p.os = d->out_scale;
    p.relu = d->relu;
    p.zero = (const half_t*)d->zero_page;
    p.in2 = (const half_t*)d->in2;
    p.cin2_shift = d->in2 ? __builtin_ctz((unsigned)d->cin2) : 0;
    p.kt2 = d->in2 ? d->ntaps * d->cin / CBK : 0x7fffffff;
    p.nvar = n;
    for (int i = 0; i < n; ++i) {
        p.var[i].taps = descs[i].taps; p.var[i].w = (const half_t*)descs[i].w;
        p.var[i].pz = descs[i].out_off_z; p.var[i].py = descs[i].out_off_y; p.var[i].px = descs[i].out_off_x;
    }
    int64_t m; int splits;
    conv_shape(d, n, &m, &p.tiles_n, &splits);
    p.M = (int)m;
    const size_t need = splits > 1 ? (size_t)splits * n * m * d->cout * sizeof(float) : 0;
    if (need > 0 && (workspace == nullptr || workspace_bytes < need)) splits = 1;   // no scratch: unsplit launch
    p.splits = splits;
    p.slabs = splits > 1 ? (float*)workspace : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(ceil_div(m, 128) * p.tiles_n), (unsigned)n, (unsigned)splits);
    // C_out <= 64, long K (the k4 s2 down-convolutions): three stages, the gather of K tile kt+2 in flight behind the
    // barrier of kt (enc.3 141 -> 112 us); short K and the 128-wide tile measured equal or slower with deeper rings
    if (d->cout <= 64) {
        if (d->kpad / CBK >= 32) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 64, 64, 3>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv3d_igemm_kernel<128, 64, 64, 2>), grid, dim3(256), 0, s, p);
    } else {
        if (g_igemm_abl == 1) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2, 1>), grid, dim3(256), 0, s, p);
        else if (g_igemm_abl == 2) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2, 2>), grid, dim3(256), 0, s, p);
        else if (g_igemm_abl == 3) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2, 3>), grid, dim3(256), 0, s, p);
        else if (g_igemm_abl == 4) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2, 4>), grid, dim3(256), 0, s, p);
        else if (g_igemm_abl == 7) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2, 7>), grid, dim3(256), 0, s, p);
        else if (g_igemm_abl == 8) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2, 8>), grid, dim3(256), 0, s, p);
        else if (g_igemm_abl == 15) hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2, 15>), grid, dim3(256), 0, s, p);
        else
        hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128, 64, 2>), grid, dim3(256), 0, s, p);
    }
    PCD_CHECK_LAUNCH();
    if (splits > 1) {
        const int64_t per = m * (d->cout / 8);                       // < 2^31: conv_check bounds the rows, cout / 8 <= 2^12 here
        PCD_CHECK_ARG(per <= 0x7fffffff);
        hipLaunchKernelGGL(conv3d_finish_kernel, dim3((unsigned)ceil_div(per, 256), (unsigned)n), dim3(256), 0, s, p);
        PCD_CHECK_LAUNCH();
    }
    return PCD_OK;
}

extern "C" int pcd_conv3d_f16(const pcd_conv3d_desc_t* d, void* stream) {
    PCD_CHECK_ARG(d != nullptr);
    return pcd_conv3d_f16_multi(d, 1, nullptr, 0, stream);
}

static int g_halo_tall = 1;          // tuning / testing hook (pcd_conv3d_config)
static int g_first8 = 1;      // the first layer (Conv3d 1 -> 32): 8 x 8 x 8 tiles where the grid divides (default), 0 = 4 x 4 x 8 tiles (pcd_conv3d_config + 512)
static int g_last_roll = 1;   // the per-tap form with every input slice read once by a wave that owns four output slices (default); 0 = a wave per output slice (pcd_conv3d_config + 16384)
static int g_last_abl = 0;    // timing ablations of conv3d_last_taps_kernel (pcd_conv3d_config + 128 / + 256; outputs wrong)
static int g_last8 = 3;       // the last layer (Conv3d 32 -> 1 + sigmoid): 3 = per-tap partial products (default), 2 = 8 x 8 x 8 blocks with one MFMA per tap and 16 voxels
                              // (pcd_conv3d_config + 24), 1 = 8 x 8 x 8 on the VALU (+ 16), 0 = 4 x 4 x 8 blocks (+ 8)

extern "C" int pcd_conv3d_config(int tall_halo_tiles) {
    PCD_CHECK_ARG(tall_halo_tiles >= 0 && (tall_halo_tiles & 7) <= 2 && tall_halo_tiles < 262144);
    g_wreg_abl = (tall_halo_tiles >> 15) & 7;
    g_last_roll = (tall_halo_tiles & 16384) ? 0 : 1;
    g_igemm_abl = (tall_halo_tiles >> 10) & 15;
    g_first8 = (tall_halo_tiles & 512) ? 0 : 1;
    g_last_abl = (tall_halo_tiles >> 7) & 3;
    g_split_target = (tall_halo_tiles & 96) == 96 ? 1024 : (tall_halo_tiles & 32) ? 768 : ((tall_halo_tiles & 64) ? 384 : 512);
    g_halo_tall = tall_halo_tiles & 7;
    g_last8 = (tall_halo_tiles & 24) == 24 ? 2 : ((tall_halo_tiles & 8) ? 0 : ((tall_halo_tiles & 16) ? 1 : 3));
    return PCD_OK;
}

static bool halo_supported(const pcd_conv3d_desc_t* d) {
    return d->ntaps == 27 && d->stride == 1 && d->out_scale == 1 && (d->cin == 32 || d->cin == 64) &&
           d->cout % 8 == 0 && d->kpad >= 27 * d->cin &&
           d->rows_d == d->in_d && d->rows_h == d->in_h && d->rows_w == d->in_w && d->out_d == d->in_d &&
           d->out_h == d->in_h && d->out_w == d->in_w && d->in_d % HTZ == 0 && d->in_h % HTY == 0 &&
           d->in_w % HTX == 0 && d->out_off_z == 0 && d->out_off_y == 0 && d->out_off_x == 0 &&
           (d->in2 == nullptr || (d->cin == 64 && d->cout % 64 == 0 && d->cin2 == 32));
}

extern "C" int pcd_conv3d_k3s1_supported(const pcd_conv3d_desc_t* d) {
    return d != nullptr && conv_check(d) == PCD_OK && halo_supported(d) ? 1 : 0;
}

extern "C" int pcd_conv3d_k3s1_f16(const pcd_conv3d_desc_t* d, void* stream) {
    PCD_CHECK_ARG(d != nullptr);
    const int rc = conv_check(d);
    if (rc != PCD_OK) return rc;
    PCD_CHECK_ARG(halo_supported(d));
    HaloParams p{};
    p.in = (const half_t*)d->in; p.B = d->batch; p.D = d->in_d; p.H = d->in_h; p.W = d->in_w;
    p.w = (const half_t*)d->w; p.kpad = d->kpad; p.bias = d->bias; p.resid = (const half_t*)d->resid;
    p.out = (half_t*)d->out; p.Cout = d->cout; p.relu = d->relu;
    p.in2 = (const half_t*)d->in2;
    const int bn = d->cout <= 32 ? 32 : 64;
    p.tiles_n = (int)ceil_div(d->cout, bn);
    // 32 -> 32: 256-row workgroups (4 x 8 x 8 voxels, eight waves) where they still give every CU two rounds of work
    const int64_t blocks256 = (int64_t)d->batch * (d->in_d / HTZ) * (d->in_h / 8) * (d->in_w / HTX) * p.tiles_n;
    const bool tall = d->cin == 32 && bn == 32 && d->in_h % 8 == 0 &&
                      (g_halo_tall == 2 || (g_halo_tall == 1 && blocks256 >= 512));
    const int ty_rows = tall ? 8 : HTY;
    p.tz = d->in_d / HTZ; p.ty = d->in_h / ty_rows; p.tx = d->in_w / HTX;
    const int64_t blocks = (int64_t)d->batch * p.tz * p.ty * p.tx * p.tiles_n;
    PCD_CHECK_ARG(blocks <= 0x7fffffff);
    p.nblocks = (int)blocks;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), blk(tall ? 512 : 256);
    // weight stage = one tap (three taps for 32 -> 32, where a tap is only 4 MFMAs per wave); 4 waves: the
    // 2-wave / 64 x 64 wave-tile form (fewer LDS reads per MFMA, but one wave per SIMD) measured 345 vs 304 us
    if (tall) hipLaunchKernelGGL((conv3d_halo_kernel<32, 32, 3, 4, 8, 8>), grid, blk, 0, s, p);
    else if (d->in2 != nullptr) hipLaunchKernelGGL((conv3d_halo_kernel<64, 64, 1, 4, 4, 4, 32>), grid, blk, 0, s, p);
    else if (d->cin == 64 && bn == 64) hipLaunchKernelGGL((conv3d_halo_kernel<64, 64, 1, 4, 4>), grid, blk, 0, s, p);
    else if (d->cin == 64) hipLaunchKernelGGL((conv3d_halo_kernel<64, 32, 1, 4, 4>), grid, blk, 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((conv3d_halo_kernel<32, 64, 1, 4, 4>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((conv3d_halo_kernel<32, 32, 3, 4, 4>), grid, blk, 0, s, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// wave columns of a tile: C_out = 32 layers run 32-wide tiles, C_in = 128 layers 128-wide tiles on eight waves, everything else 64-wide tiles on four
static int wreg_nwc(int cin, int cout) { return cin == 128 ? 4 : (cout == 32 ? 1 : 2); }
static bool wreg_shape_ok(int cin, int cout) {
    return cout > 0 && (((cin == 64 || cin == 32) && (cout == 32 || cout % 64 == 0)) || (cin == 128 && cout % 128 == 0));
}
static int wreg_stages(int cin) { return 27 * (cin > 64 ? cin / 64 : 1) + 1; }

extern "C" size_t pcd_conv3d_wfrag_bytes(int cin, int cout) {
    return wreg_shape_ok(cin, cout) ? (size_t)wreg_stages(cin) * (cin > 64 ? 64 : cin) * cout * sizeof(half_t) : 0;
}

extern "C" int pcd_conv3d_pack_wfrag(const void* w, int kpad, int cin, int cout, int cin2, void* wfrag, void* stream) {
    PCD_CHECK_ARG(w && wfrag && wreg_shape_ok(cin, cout) && (cin2 == 0 || (cin >= 64 && (cin2 == 32 || cin2 == 64))));
    PCD_CHECK_ARG(kpad >= 27 * cin + cin2 && kpad % 8 == 0);
    const int nwc = wreg_nwc(cin, cout);
    const int total = (cout / (32 * nwc)) * wreg_stages(cin) * nwc * ((cin > 64 ? 64 : cin) / 32) * 2 * 64;
    hipLaunchKernelGGL(conv3d_pack_wfrag_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)w, kpad, cin, cout,
                       cin2, nwc, (half_t*)wfrag);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

static bool wreg_supported(const pcd_conv3d_desc_t* d) {
    return d->ntaps == 27 && d->stride == 1 && d->out_scale == 1 && wreg_shape_ok(d->cin, d->cout) &&
           (d->in2 == nullptr || (d->cin >= 64 && (d->cin2 == 32 || d->cin2 == 64))) &&
           d->rows_d == d->in_d && d->rows_h == d->in_h && d->rows_w == d->in_w && d->out_d == d->in_d && d->out_h == d->in_h && d->out_w == d->in_w &&
           d->in_d % HTZ == 0 && d->in_h % 8 == 0 && d->in_w % HTX == 0 && d->out_off_z == 0 && d->out_off_y == 0 && d->out_off_x == 0;
}

extern "C" int pcd_conv3d_k3s1_wreg_supported(const pcd_conv3d_desc_t* d) { return d != nullptr && conv_check(d) == PCD_OK && wreg_supported(d) ? 1 : 0; }

extern "C" int pcd_conv3d_k3s1_wreg_f16(const pcd_conv3d_desc_t* d, const void* wfrag, void* stream) {
    PCD_CHECK_ARG(d != nullptr && wfrag != nullptr);
    const int rc = conv_check(d);
    if (rc != PCD_OK) return rc;
    PCD_CHECK_ARG(wreg_supported(d));
    HaloWregParams p{};
    p.in = (const half_t*)d->in; p.B = d->batch; p.D = d->in_d; p.H = d->in_h; p.W = d->in_w;
    p.in2 = (const half_t*)d->in2; p.cin2 = d->in2 ? d->cin2 : 0;
    p.wfrag = (const half_t*)wfrag; p.bias = d->bias; p.resid = (const half_t*)d->resid;
    p.out = (half_t*)d->out; p.Cout = d->cout; p.relu = d->relu;
    const int nwc = wreg_nwc(d->cin, d->cout);
    p.tiles_n = d->cout / (32 * nwc); p.tz = d->in_d / HTZ; p.ty = d->in_h / 8; p.tx = d->in_w / HTX;
    const int64_t blocks = (int64_t)d->batch * p.tz * p.ty * p.tx * p.tiles_n;
    PCD_CHECK_ARG(blocks <= 0x7fffffff);
    p.nblocks = (int)blocks;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks);
    if (d->cin == 128) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<128, 4>), grid, dim3(512), 0, s, p);              // eight waves, one workgroup per CU
    else if (d->cin == 64 && nwc == 2 && g_wreg_abl == 1) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<64, 2, 2, 1>), grid, dim3(256), 0, s, p);
    else if (d->cin == 64 && nwc == 2 && g_wreg_abl == 2) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<64, 2, 2, 2>), grid, dim3(256), 0, s, p);
    else if (d->cin == 64 && nwc == 2 && g_wreg_abl == 4) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<64, 2, 2, 4>), grid, dim3(256), 0, s, p);
    else if (d->cin == 64 && nwc == 2 && g_wreg_abl == 6) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<64, 2, 2, 6>), grid, dim3(256), 0, s, p);
    else if (d->cin == 64 && nwc == 2) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<64, 2>), grid, dim3(256), 0, s, p);
    else if (d->cin == 64) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<64, 1, 4>), grid, dim3(256), 0, s, p);     // C_out 32: four waves of 64 voxels
    else if (nwc == 2) hipLaunchKernelGGL((conv3d_halo_wreg_kernel<32, 2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv3d_halo_wreg_kernel<32, 1>), grid, dim3(128), 0, s, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_conv3d_k4s2_halo_supported(int batch, int d, int h, int w, int cin, int cout, int kpad) {
    return batch > 0 && cin == 64 && cout == 64 && kpad >= 64 * 64 && kpad % 8 == 0 && d > 0 && h > 0 && w > 0 && d % (2 * HTZ) == 0 &&
           h % (2 * HTY) == 0 && w % (2 * HTX) == 0 && d <= 64 && h <= 64 && w <= 64 && ((int64_t)batch + 1) * d * h * w * cin <= 0x7fffffff ? 1 : 0;      // (+ 1: the kernel forms offsets one plane past the end)
}

extern "C" int pcd_conv3d_k4s2_halo_f16(const void* in, int batch, int d, int h, int w, int cin, const void* wgt, int kpad, const float* bias,
                                        int relu, int cout, void* out, void* stream) {
    PCD_CHECK_ARG(in && wgt && out);
    PCD_CHECK_ARG(pcd_conv3d_k4s2_halo_supported(batch, d, h, w, cin, cout, kpad));
    ConvS2HaloParams p{};
    p.in = (const half_t*)in; p.B = batch; p.D = d; p.H = h; p.W = w;
    p.w = (const half_t*)wgt; p.kpad = kpad; p.bias = bias; p.out = (half_t*)out; p.relu = relu;
    p.tz = d / 2 / HTZ; p.ty = h / 2 / HTY; p.tx = w / 2 / HTX;
    const int64_t blocks = (int64_t)batch * p.tz * p.ty * p.tx;
    PCD_CHECK_ARG(blocks <= 0x7fffffff);
    p.nblocks = (int)blocks;
    hipLaunchKernelGGL(conv3d_k4s2_halo_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_convt3d_k4s2_halo_supported(int batch, int d, int h, int w, int cin, int cout) {
    return batch > 0 && cin == 128 && cout == 64 && d > 0 && h > 0 && w > 0 && d % HTZ == 0 && h % HTY == 0 && w % HTX == 0 &&
           (int64_t)batch * d * h * w * 8 * cout <= 0x7fffffff ? 1 : 0;
}

extern "C" int pcd_convt3d_k4s2_halo_f16(const void* in, int batch, int d, int h, int w, int cin, const void* const* w8, const float* bias,
                                         int cout, void* out, void* stream) {
    PCD_CHECK_ARG(in && w8 && out);
    PCD_CHECK_ARG(pcd_convt3d_k4s2_halo_supported(batch, d, h, w, cin, cout));
    ConvTHaloParams p{};
    p.in = (const half_t*)in; p.B = batch; p.D = d; p.H = h; p.W = w;
    for (int k = 0; k < 8; ++k) {
        PCD_CHECK_ARG(w8[k] != nullptr);
        p.w[k] = (const half_t*)w8[k];
    }
    p.bias = bias; p.out = (half_t*)out;
    p.tz = d / HTZ; p.ty = h / HTY; p.tx = w / HTX;
    const int64_t blocks = (int64_t)batch * p.tz * p.ty * p.tx;
    PCD_CHECK_ARG(blocks <= 0x7fffffff);
    p.nblocks = (int)blocks;
    hipLaunchKernelGGL(convT3d_halo_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_conv3d_first(const float* x, int batch, int d, int h, int w, int stride, const float* wgt,
                                const float* bias, int cout, void* out, void* stream) {
    PCD_CHECK_ARG(x && wgt && bias && out && batch > 0 && d > 0 && h > 0 && w > 0 && cout > 0 && cout % 8 == 0);
    PCD_CHECK_ARG(stride == 1 || stride == 2);
    const int64_t ovox = (int64_t)((d - 1) / stride + 1) * ((h - 1) / stride + 1) * ((w - 1) / stride + 1);
    if (cout == 32 && stride == 1 && g_first8 && d % 8 == 0 && h % 8 == 0 && w % 8 == 0 && (int64_t)batch * ovox / 512 <= 0x7fffffff) {
        const int64_t tiles = (int64_t)batch * ovox / 512;
        hipLaunchKernelGGL((conv3d_first32_mfma_kernel<8, 8>), dim3((unsigned)(tiles < 2048 ? tiles : 2048)), dim3(256), 0,
                           (hipStream_t)stream, x, batch, d, h, w, wgt, bias, (half_t*)out, d / 8, h / 8, w / 8, (int)tiles);
        PCD_CHECK_LAUNCH();
        return PCD_OK;
    }
    if (cout == 32 && stride == 1 && d % HTZ == 0 && h % HTY == 0 && w % HTX == 0 && (int64_t)batch * ovox / 128 <= 0x7fffffff) {
        const int64_t tiles = (int64_t)batch * ovox / 128;
        hipLaunchKernelGGL((conv3d_first32_mfma_kernel<4, 4>), dim3((unsigned)(tiles < 2048 ? tiles : 2048)), dim3(256), 0,
                           (hipStream_t)stream, x, batch, d, h, w, wgt, bias, (half_t*)out, d / HTZ, h / HTY, w / HTX,
                           (int)tiles);
        PCD_CHECK_LAUNCH();
        return PCD_OK;
    }
    if (cout == 32) {
        hipLaunchKernelGGL(conv3d_first32_kernel, dim3((unsigned)ceil_div((int64_t)batch * ovox, 256)), dim3(256), 0,
                           (hipStream_t)stream, x, batch, d, h, w, stride, wgt, bias, (half_t*)out);
        PCD_CHECK_LAUNCH();
        return PCD_OK;
    }
    const int64_t total = (int64_t)batch * ovox * (cout / 8);
    hipLaunchKernelGGL(conv3d_first_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256),
                       (size_t)cout * 28 * sizeof(float), (hipStream_t)stream, x, batch, d, h, w, stride, wgt, bias,
                       cout, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_conv3d_last_sigmoid(const void* in, int batch, int d, int h, int w, int cin, const float* wgt,
                                       float bias, float* out, void* stream) {
    PCD_CHECK_ARG(in && wgt && out && batch > 0 && d > 0 && h > 0 && w > 0);
    PCD_CHECK_ARG(cin == 32);
    const int64_t total = (int64_t)batch * d * h * w;
    if (g_last8 != 0 && d % 8 == 0 && h % 8 == 0 && w % 8 == 0 && total / 512 <= 0x7fffffff) {
        hipLaunchKernelGGL(conv3d_last_halo8_kernel<8>, dim3((unsigned)(total / 512)), dim3(512), 0, (hipStream_t)stream,
                           (const half_t*)in, batch, d, h, w, wgt, bias, out, d / 8, h / 8, w / 8);
    } else if (d % HTZ == 0 && h % HTY == 0 && w % HTX == 0 && total / 128 <= 0x7fffffff) {
        hipLaunchKernelGGL(conv3d_last_halo_kernel, dim3((unsigned)(total / 128)), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)in, batch, d, h, w, wgt, bias, out, d / HTZ, h / HTY, w / HTX);
    } else {
        hipLaunchKernelGGL((conv3d_last_kernel<32>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, (const half_t*)in, batch, d, h, w, wgt, bias, out);
    }
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" size_t pcd_conv3d_last_packed_bytes(void) { return (size_t)27 * 64 * 8 * sizeof(half_t); }

extern "C" int pcd_conv3d_last_pack(const float* wgt, void* wfrag, void* stream) {
    PCD_CHECK_ARG(wgt && wfrag);
    hipLaunchKernelGGL(conv3d_last_pack_kernel, dim3(27), dim3(64), 0, (hipStream_t)stream, wgt, (half_t*)wfrag);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// the same layer from pcd_conv3d_last_pack's copy of the weights, on the matrix pipe (d, h, w multiples of 8; otherwise, or with pcd_conv3d_config(+ 8 | + 16),
// pcd_conv3d_last_sigmoid's kernels run from the fp32 weights)
extern "C" int pcd_conv3d_last_sigmoid_packed(const void* in, int batch, int d, int h, int w, int cin, const float* wgt, const void* wfrag, float bias, float* out,
                                              void* stream) {
    PCD_CHECK_ARG(in && wgt && out && batch > 0 && d > 0 && h > 0 && w > 0 && cin == 32);
    const int64_t total = (int64_t)batch * d * h * w;
    if (wfrag != nullptr && g_last8 == 3 && g_last_roll && d % 4 == 0 && h % 8 == 0 && w % 8 == 0 && total / 256 <= 0x7fffffff) {
        const int64_t nwork = total / 256;                          // (sample, 4 output slices, 8 x 8 tile) units, one per wave
        hipLaunchKernelGGL(conv3d_last_roll_kernel<4>, dim3((unsigned)ceil_div(nwork, 4)), dim3(256), 0, (hipStream_t)stream, (const half_t*)in, batch, d, h, w,
                           (const half_t*)wfrag, bias, out, d / 4, h / 8, w / 8, (int)nwork);
        PCD_CHECK_LAUNCH();
        return PCD_OK;
    }
    if (wfrag != nullptr && g_last8 == 3 && d % 4 == 0 && h % 8 == 0 && w % 8 == 0 && total / 256 <= 0x7fffffff) {
        if (g_last_abl == 1) hipLaunchKernelGGL((conv3d_last_taps_kernel<4, 1>), dim3((unsigned)(total / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)in, batch, d, h, w, (const half_t*)wfrag, bias, out, d / 4, h / 8, w / 8);
        else if (g_last_abl == 2) hipLaunchKernelGGL((conv3d_last_taps_kernel<4, 2>), dim3((unsigned)(total / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)in, batch, d, h, w, (const half_t*)wfrag, bias, out, d / 4, h / 8, w / 8);
        else
        hipLaunchKernelGGL((conv3d_last_taps_kernel<4>), dim3((unsigned)(total / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)in, batch, d, h, w, (const half_t*)wfrag, bias, out,
                           d / 4, h / 8, w / 8);
        PCD_CHECK_LAUNCH();
        return PCD_OK;
    }
    if (wfrag == nullptr || g_last8 != 2 || d % 8 || h % 8 || w % 8 || total / 512 > 0x7fffffff)
        return pcd_conv3d_last_sigmoid(in, batch, d, h, w, cin, wgt, bias, out, stream);
    hipLaunchKernelGGL(conv3d_last_mfma8_kernel, dim3((unsigned)(total / 512)), dim3(512), 0, (hipStream_t)stream, (const half_t*)in, batch, d, h, w,
                       (const half_t*)wfrag, bias, out, d / 8, h / 8, w / 8);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_convt3d_last_sigmoid(const void* in, int batch, int d, int h, int w, int cin, const float* wgt,
                                        float bias, float* out, void* stream) {
    PCD_CHECK_ARG(in && wgt && out && batch > 0 && d > 0 && h > 0 && w > 0);
    PCD_CHECK_ARG(cin == 32);
    const int64_t total = (int64_t)batch * d * h * w * 8;
    hipLaunchKernelGGL((convT3d_last_kernel<32>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const half_t*)in, batch, d, h, w, wgt, bias, out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
