// GEMM / implicit-GEMM convolution, phase-split ("ping-pong") main loop for the 256-row tiles.
//
// Same operands, LDS image, MFMA roles, persistent XCD-aware tile walk and epilogue as gemm_dma.hip; what changes is
// how a K-tile is executed by the 8 waves of a block.  gemm_dma.hip runs all waves in lock step: everybody issues the
// operand DMA, everybody reads fragments, everybody issues MFMAs, `s_waitcnt vmcnt(0)` + `__syncthreads()` per K-step -
// so the two waves of a SIMD want the matrix pipe at the same time and leave it idle at the same time (measured: a
// K-step costs MFMA + DMA issue + barrier, 2.05 us against 1.07 us of MFMA work, profiles/r01_gemm_experiments.md).
//
// Here the two waves of a SIMD (wave w and w + 4) belong to two GROUPS that run one SLOT apart:
//     slot        4S        4S+1      4S+2      4S+3      4S+4
//     group 0     L0(S)     M0(S)     L1(S)     M1(S)     L0(S+1)  ...        S = K-tile, kk = its 32-deep half
//     group 1     M1(S-1)   L0(S)     M0(S)     L1(S)     M1(S)    ...
//   L slot: ds_read the fragments of one K half (4 activation + NF weight fragments) + issue this wave's DMA pieces;
//   M slot: the 4 x NF MFMAs of that half, nothing else.  Slots are separated by raw `s_barrier`s (never __syncthreads:
//   that would drain the LDS-DMA queue), so on every SIMD one wave feeds the matrix pipe while the other one sits in its
//   LDS reads and DMA issue.  Group g owns the activation rows 128 g .. 128 g + 127 of the tile (its waves fetch exactly
//   those rows and only they read them); both groups read all weight rows.
//
// Two LDS stages per operand, DMA of K-tile S+1 issued during K-tile S, never drained inside the loop:
//   activation rows (own):  group 0 issues in L0(S) [slot 4S],   waits at the end of M1(S) [4S+3]  -> 3+ slots of flight
//                           group 1 issues in L0(S) [slot 4S+1], waits at the end of M1(S) [4S+4]  -> 3+ slots
//   weight rows (shared):   group 1 issues in L0(S) [slot 4S+1] BEFORE its activation pieces and waits `vmcnt(XP)` at the
//                           end of L1(S) [4S+3]; group 0 issues in L1(S) [slot 4S+2] and waits at the end of M1(S) [4S+3];
//                           first reader of W(S+1) is group 0's L0(S+1) [slot 4S+4], behind the barrier that ends slot 4S+3.
//   WAR: stage (S+1)&1 held K-tile S-1.  Own activation rows: last read in L1(S-1), returned by the lgkmcnt wait of
//   M1(S-1), which precedes L0(S) of the same group.  Weights: last reader is group 1's L1(S-1) [slot 4S-1], returned in
//   its M1(S-1) [slot 4S]; the earliest weight DMA is group 1's L0(S) [4S+1] / group 0's L1(S) [4S+2].  (Group 0 must
//   NOT issue weight pieces in its L0(S): slot 4S is where group 1 still waits for its last W(S-1) fragments.)
//
// Tile boundary: the groups re-align for the epilogue (group 0 waits one barrier for group 1's last M slot, both run their
// epilogues at the same time - otherwise every barrier would serialise the two 10 us epilogues - and group 1 re-creates
// its one-slot lag with one extra barrier afterwards).  The DMA of the next tile's first K-tile is already in flight.
#include "gemm_epilogue.h"

using namespace vcxgemm;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
[[maybe_unused]] constexpr unsigned OOB = 0xFFFFFFFFu;

template <int TBM_, int TBN_>
struct PPCfg {
    static constexpr int TBM = TBM_, TBN = TBN_, NWM = 4, NWN = 2;
    static constexpr int THREADS = 512;
    static constexpr int MF = TBM / NWM / 16;      // 16-row activation fragments per wave (4)
    static constexpr int NF = TBN / NWN / 16;      // 16-col weight fragments per wave (8 / 10)
    static constexpr int XP = TBM / 2 / 32;        // DMA pieces per thread for its group's 128 activation rows (4)
    static constexpr int WP = TBN / 64;            // DMA pieces per thread for the weight rows (4 / 5)
    static constexpr size_t STAGES = (size_t)2 * (TBM + TBN) * BK * sizeof(half_t);
    static constexpr size_t SMEM = STAGES + (size_t)TBN * NWM * sizeof(float);   // + one bias strip per wave (epilogue)
};

#define VCX_WAIT_VM(n) __builtin_amdgcn_s_waitcnt((((n) >> 4) << 14) | 0x0f70 | ((n) & 15))
#define VCX_SLOT_BARRIER()                     \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

template <class Cfg, bool CONV, bool GEGLU, bool OUT_F32>
__global__ void __launch_bounds__(Cfg::THREADS, 2) gemm_pp_kernel(GemmArgs p, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the body uses device-only types)
    constexpr int TBM = Cfg::TBM, BN = Cfg::TBN;
    constexpr int NFRAG = Cfg::NF, MFRAG = Cfg::MF, XP = Cfg::XP, WP = Cfg::WP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* sX = reinterpret_cast<half_t*>(smem_raw);              // [2][TBM*BK]
    half_t* sW = sX + 2 * TBM * BK;                                 // [2][BN*BK]

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.W), 0, (int)w_bytes, 0x00020000);

    const int ntiles = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp = wave >> 2, wq = wave & 3;           // group = which wave of the SIMD; position inside the group
    const int wm = 2 * grp + (wq & 1), wn = wq >> 1;    // 4 (M) x 2 (N) wave grid, rows 128 g .. 128 g + 127 to group g
    const int chunk = tid & 7;                          // LDS chunk position inside the 128-byte row
    const int r0x = grp * 128 + ((tid & 255) >> 3);     // tile row of this thread's first activation piece (own rows)
    const int r0w = tid >> 3;                           // tile row of its first weight piece

    // ---- per-thread source offsets (bytes) of the tile being loaded
    unsigned xoff[XP];           // activation rows; OOB when the row is beyond M (linear mode)
    unsigned xmask[XP];          // conv: bit t set <=> tap t of this row is inside the image
    unsigned woff[WP];           // weight rows; OOB when beyond N
    int tap = 0, ci0 = 0;        // conv K walker (block-uniform): k = tap*cin + ci0
    unsigned tap_off = 0;        // conv: byte offset of (tap, ci0) relative to the row's (ky=0, kx=0, c=0) pixel
    int tky = 0, tkx = 0;
    auto init_load = [&](int t) {
        int tile_m, tile_n;
        tile_coords(t, ntiles, p.tiles_n, tile_m, tile_n);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int r = r0x + 32 * i;
            const int m = p.m_begin + tile_m * TBM + r;
            const unsigned csrc = (unsigned)(chunk ^ ((r >> 1) & 7)) * 16u;   // source chunk that lands at position `chunk`
            if (CONV) {
                const int hw = p.out_h * p.out_w;
                const int mm = m < p.M ? m : 0;
                const int img = mm / hw;
                const int rem = mm - img * hw;
                const int oy = rem / p.out_w;
                const int ox = rem - oy * p.out_w;
                const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
                const long long pix0 = ((long long)img * p.in_h + (iy0 >> p.ups)) * p.in_w + (ix0 >> p.ups);   // may be negative at the border
                xoff[i] = (unsigned)(pix0 * p.lda * 2) + csrc;                            // wraps; valid taps un-wrap it
                unsigned mask = 0;
                if (m < p.M) {
                    const int lim_h = p.in_h << p.ups, lim_w = p.in_w << p.ups;
                    for (int ky = 0; ky < p.kh; ++ky)
                        for (int kx = 0; kx < p.kw; ++kx) {
                            const int iy = iy0 + ky, ix = ix0 + kx;
                            if (iy >= 0 && iy < lim_h && ix >= 0 && ix < lim_w) mask |= 1u << (ky * p.kw + kx);
                        }
                }
                xmask[i] = mask | ((unsigned)(iy0 & 1) << 31) | ((unsigned)(ix0 & 1) << 30);   // parity bits for the fused 2x upsampling
            } else {
                xoff[i] = m < p.M ? (unsigned)((long long)m * p.lda * 2) + csrc : OOB;
                xmask[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int r = r0w + 64 * i;
            const int n = tile_n * BN + r;
            const unsigned csrc = (unsigned)(chunk ^ ((r >> 1) & 7)) * 16u;
            woff[i] = n < p.N ? (unsigned)((long long)n * p.ldw * 2) + csrc : OOB;
        }
        tap = 0; ci0 = 0; tap_off = 0; tky = 0; tkx = 0;
    };

    // DMA of this thread's own activation rows of load K-tile kt into LDS stage `buf` (and advance the conv K walker)
    auto issue_x = [&](int kt, int buf) {
        half_t* dx = sX + buf * TBM * BK + (grp * 128 + wq * 8) * BK;
        if (CONV) {
            if (p.ups) {
                const unsigned cb = (unsigned)ci0 * 2u, rowb = (unsigned)(p.in_w * (int)p.lda * 2), pixb = (unsigned)((int)p.lda * 2);
                const unsigned y0 = (unsigned)(tky >> 1) * rowb, y1 = (unsigned)((tky + 1) >> 1) * rowb;
                const unsigned x0 = (unsigned)(tkx >> 1) * pixb, x1 = (unsigned)((tkx + 1) >> 1) * pixb;
#pragma unroll
                for (int i = 0; i < XP; ++i) {
                    const unsigned ok = (xmask[i] >> tap) & 1u;
                    const unsigned oy_ = (xmask[i] >> 31) ? y1 : y0, ox_ = ((xmask[i] >> 30) & 1u) ? x1 : x0;
                    const unsigned v = ok ? xoff[i] + oy_ + ox_ + cb : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + 32 * i * BK), 16, v, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < XP; ++i) {
                    const unsigned ok = (xmask[i] >> tap) & 1u;
                    const unsigned v = ok ? xoff[i] + tap_off : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + 32 * i * BK), 16, v, 0, 0, 0);
                }
            }
            if (p.flags & VCX_GEMM_CONV_SLABK) {         // taps inside a 64-channel slab
                ++tap;
                if (++tkx == p.kw) { tkx = 0; ++tky; }
                if (tap == p.kh * p.kw) { tap = 0; tkx = 0; tky = 0; ci0 += BK; }
                tap_off = (unsigned)((tky * p.in_w + tkx) * (int)p.lda * 2 + ci0 * 2);
            } else {                                     // channel slabs inside a tap
                ci0 += BK;
                tap_off += BK * 2;
                if (ci0 == p.cin) {
                    ci0 = 0;
                    ++tap;
                    if (++tkx == p.kw) { tkx = 0; ++tky; }
                    tap_off = (unsigned)((tky * p.in_w + tkx) * (int)p.lda * 2);
                }
            }
        } else {
            const unsigned soff = (unsigned)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < XP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + 32 * i * BK), 16, xoff[i], soff, 0, 0);
        }
    };
    auto issue_w = [&](int kt, int buf) {
        half_t* dw = sW + buf * BN * BK + wave * 8 * BK;
        const unsigned soffw = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr_t)(dw + 64 * i * BK), 16, woff[i], soffw, 0, 0);
    };

    const int lr = lane & 15, lg = lane >> 4;
    constexpr int WM = TBM / Cfg::NWM, WN = BN / Cfg::NWN;   // wave tile

    f4 acc[NFRAG][MFRAG];
#pragma unroll
    for (int a = 0; a < NFRAG; ++a)
#pragma unroll
        for (int b = 0; b < MFRAG; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    const bool prio = p.tune & 2;
    int ltile = blockIdx.x, lkt = 0;
    int ctile = blockIdx.x, ckt = 0;
    int tile_m, tile_n;
    tile_coords(ctile, ntiles, p.tiles_n, tile_m, tile_n);
    init_load(ltile);
    issue_w(0, 0);
    issue_x(0, 0);
    VCX_WAIT_VM(0);                 // first K-tile landed in LDS
    VCX_SLOT_BARRIER();
    if (grp) VCX_SLOT_BARRIER();    // group 1 runs one slot behind group 0
    int cur = 0;
    for (;;) {
        if (++lkt == nk) {
            lkt = 0;
            ltile += G;
            if (ltile < ntiles) init_load(ltile);
        }
        const bool more = ltile < ntiles;
        const half_t* cx = sX + cur * TBM * BK;
        const half_t* cw = sW + cur * BN * BK;
        h8 xf[MFRAG], wf[NFRAG];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // ---------------- L slot: fragments of this K half + this wave's share of the next K-tile's DMA
#pragma unroll
            for (int b = 0; b < MFRAG; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * WM + b * 16 + lr, kk * 4 + lg));
#pragma unroll
            for (int a = 0; a < NFRAG; ++a) wf[a] = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + a * 16 + lr, kk * 4 + lg));
            if (kk == 0) {
                if (more) {
                    if (grp) issue_w(lkt, cur ^ 1);        // group 1: weights first (they are waited for first) ...
                    issue_x(lkt, cur ^ 1);                 // ... then the own activation rows
                }
            } else {
                if (grp) {
                    if (more) VCX_WAIT_VM(XP); else VCX_WAIT_VM(0);   // group 1's weight pieces landed; its activation pieces may fly
                } else if (more) {
                    issue_w(lkt, cur ^ 1);                 // group 0: weights one slot after group 1 released the stage
                }
            }
            VCX_SLOT_BARRIER();
            // ---------------- M slot
            if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < NFRAG; ++a)
#pragma unroll
                for (int b = 0; b < MFRAG; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            if (prio) __builtin_amdgcn_s_setprio(0);
            if (kk == 1) VCX_WAIT_VM(0);                   // everything this wave issued for the next K-tile has landed
            VCX_SLOT_BARRIER();
        }
        cur ^= 1;
        if (++ckt == nk) {
            if (!grp) VCX_SLOT_BARRIER();                  // group 0 waits for group 1's last M slot: both epilogues run together
            float* sB = reinterpret_cast<float*>(smem_raw + Cfg::STAGES) + wave * WN;
            gemm_epilogue<Cfg, GEGLU, OUT_F32>(p, acc, tile_m, tile_n, wm, wn, lane, sB);
#pragma unroll
            for (int a = 0; a < NFRAG; ++a)
#pragma unroll
                for (int b = 0; b < MFRAG; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
            ckt = 0;
            ctile += G;
            if (ctile >= ntiles) break;
            tile_coords(ctile, ntiles, p.tiles_n, tile_m, tile_n);
            if (grp) VCX_SLOT_BARRIER();                   // group 1 falls one slot behind again
        }
    }
#endif
}

template <class Cfg, bool CONV, bool GEGLU, bool OUT_F32>
int launch(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_pp_kernel<Cfg, CONV, GEGLU, OUT_F32>;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)Cfg::SMEM, "vcx_gemm_f16(pp)")) return VCX_ELAUNCH;
    const int nb = persistent_grid(a.tiles_m * a.tiles_n, 1);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::THREADS), Cfg::SMEM, s, a, a.a_bytes, a.w_bytes);
    return vcx_check_launch("vcx_gemm_f16(pp)");
}

template <class Cfg>
int dispatch(const GemmArgs& a, bool conv, bool geglu, bool f32, hipStream_t s) {
    if (geglu) {
        if constexpr (Cfg::NF % 4 == 0) return conv ? launch<Cfg, true, true, false>(a, s) : launch<Cfg, false, true, false>(a, s);
        vcx_set_error("vcx_gemm_f16(pp): GEGLU needs whole 64-column packed blocks per wave");
        return VCX_EINVAL;
    }
    if (f32) return conv ? launch<Cfg, true, false, true>(a, s) : launch<Cfg, false, false, true>(a, s);
    return conv ? launch<Cfg, true, false, false>(a, s) : launch<Cfg, false, false, false>(a, s);
}

}  // namespace

// cfg 2: 256 x 256, cfg 3: 256 x 320 (the 256-row tile configurations of launch_dma)
int vcxgemm::launch_pp(GemmArgs& a, int cfg, bool conv, bool geglu, bool f32, hipStream_t s) {
    switch (cfg) {
        case 2: return dispatch<PPCfg<256, 256>>(a, conv, geglu, f32, s);
        case 3: return dispatch<PPCfg<256, 320>>(a, conv, geglu, f32, s);
    }
    vcx_set_error("vcx_gemm_f16(pp): tile configuration %d has no phase-split kernel", cfg);
    return VCX_EINVAL;
}
