// Attention kernels for gfx950: flash attention (head dim 64) on MFMA 32x32x16 f16, temporal
// attention over T <= 32 frames on the VALU, and an in-place row softmax.
#include "vcx_common.h"
#include "flash2.h"
#include <stdlib.h>

namespace {

// =======================================================================================
// Flash attention, d = 64.
//
// Block = 4 waves, wave w owns 32 query rows; K/V are streamed in 64-key tiles through LDS
// (K tile [64 keys][64 d], V^T tile [64 d][64 keys], both XOR-swizzled, double buffered).
//
// Per wave and key tile (all on v_mfma_f32_32x32x16_f16):
//   S^T[key, q]  = K[key, :] . Q[q, :]         (A = K fragment, B = Q fragment)
//   O^T[d, q]   += V^T[d, key] . P^T[key, q]   (A = V^T fragment, B = P fragment)
// In both products the MFMA column index is the query row (lane & 31), so the softmax
// statistics of a query live in one lane pair (lane, lane^32): the running max / sum are
// per-lane scalars and rescaling O^T is a per-lane multiply.  The S^T accumulator of lane
// (q, hi) holds keys {r&3 + 8*(r>>2) + 4*hi}; it is fed back as the B operand of the PV MFMA
// WITHOUT any cross-lane movement by reading the V^T fragment with the same key permutation
// (a sum over keys does not care about the order as long as P and V agree on it).
// =======================================================================================
[[maybe_unused]] constexpr int FK = 64;   // keys per tile
[[maybe_unused]] constexpr float FLASH_DEFER = 8.0f;   // log2 units: a tile's probabilities may reach 2^8 before the running max is moved

struct FlashArgs {
    const half_t* q;
    const half_t* k;
    const half_t* vt;
    half_t* o;
    int heads, nq, nk, kv_rows, kv_div;
    int64_t ldq, ldk, ldvt, ldo;
    float scale_log2;
    int accumulate;   // VCX_ATTN_* flag bits
    int nqb, nprob;   // query blocks per problem, problems (group x head): the 1-D grid is nqb * roundup(nprob, 8)
    // second key / value set of the DUAL kernel (O = softmax(Q K1^T) V1 + softmax(Q K2^T) V2 in one pass over Q and O)
    const half_t* k2;
    const half_t* vt2;
    int nk2, kv_rows2, kv_div2;
    int64_t ldk2, ldvt2;
};

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// QB = 32-row query blocks per wave (1 or 2).  With QB = 2 every K / V^T fragment read from LDS feeds two MFMAs, which
// halves the LDS bytes per MFMA - like the GEMM, the QB = 1 kernel is bound by LDS traffic, not by the matrix pipe.
// K / V^T tiles go HBM -> LDS by DMA (buffer_load ... lds); rows / key columns beyond nk use an out-of-range offset and
// arrive as zeros (masked to -inf before the softmax anyway).
//
// PRE (VCX_ATTN_LOG2_LOGITS): the caller folded scale * log2(e) into Q and/or K, so Q K^T is the base-2 logit itself.
// The first MFMA of a score accumulator then takes C = -m (the running max, kept in a 16-register tuple that changes only
// when the max moves) and the matrix pipe delivers s - m directly: the per-score fma of the plain path disappears
// (VALU and MFMA issue do not overlap on a SIMD - tools/ubench.hip - so every VALU instruction removed is time).
//
// DUAL: text (+) image cross-attention (attention.py:129-142) in one launch: the key / value loop runs twice, over (k, vt)
// and over (k2, vt2), each with its own softmax normalisation; the first result stays in registers (fp32) and the sum is
// rounded once.  Q is read once and O written once - run as two launches these layers are pure overhead (Q read twice,
// O written, read back and written again at 2.2-2.6 TB/s).
template <int QB, bool PRE, bool DUAL = false>
__global__ void __launch_bounds__(256, 2) flash_d64_kernel(FlashArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) half_t sK[2][64 * 64];
    __shared__ __attribute__((aligned(16))) half_t sV[2][64 * 64];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    // Workgroups go to the 8 XCDs round-robin by linear id.  All query blocks of one (group, head) problem are given ids of
    // the same residue mod 8, so the problem's K / V^T (2.4 MB at 9216 keys) is pulled into ONE XCD's L2 instead of all eight.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int prob = (slot / p.nqb) * 8 + xcd;
    if (prob >= p.nprob) return;
    const int g = prob / p.heads, h = prob % p.heads;
    const int q0 = ((slot % p.nqb) * 4 + wave) * (32 * QB);

    const half_t* qbase = p.q + ((int64_t)g * p.nq) * p.ldq + h * 64;
    // Q fragments (B operand): lane (q = lq, hi) holds Q[q][s*16 + hi*8 .. +7]
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    h8 qf[QB][4];
    bool qvalid[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qrow = q0 + b * 32 + lq;
        qvalid[b] = qrow < p.nq;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[b][s] = qvalid[b] ? *reinterpret_cast<const h8*>(qbase + (int64_t)qrow * p.ldq + s * 16 + hi * 8) : zero8;
    }

    // DUAL: normalised result of the first key set - fp32 with one query block per wave, packed fp16 with two (registers)
    typedef half_t h16v __attribute__((ext_vector_type(16)));
    f16v keep[(DUAL && QB == 1) ? QB : 1][2];
    h16v keep16[(DUAL && QB == 2) ? QB : 1][2];
#pragma unroll 1
  for (int set = 0; set < (DUAL ? 2 : 1); ++set) {
    const half_t* k_s = (DUAL && set) ? p.k2 : p.k;
    const half_t* vt_s = (DUAL && set) ? p.vt2 : p.vt;
    const int nk_s = (DUAL && set) ? p.nk2 : p.nk, kv_rows_s = (DUAL && set) ? p.kv_rows2 : p.kv_rows;
    const int kv_div_s = (DUAL && set) ? p.kv_div2 : p.kv_div;
    const int64_t ldk_s = (DUAL && set) ? p.ldk2 : p.ldk, ldvt_s = (DUAL && set) ? p.ldvt2 : p.ldvt;
    const int64_t kvrow0 = (int64_t)(g / kv_div_s) * kv_rows_s;
    const half_t* kbase = k_s + kvrow0 * ldk_s + h * 64;
    const half_t* vbase = vt_s + (int64_t)(h * 64) * ldvt_s + kvrow0;
    const unsigned k_bytes = (unsigned)(((int64_t)(nk_s - 1) * ldk_s + 64) * 2);
    const unsigned v_bytes = (unsigned)((63ll * ldvt_s + ((nk_s + 7) & ~7)) * 2);
    const __amdgpu_buffer_rsrc_t srd_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kbase), 0, (int)k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vbase), 0, (int)v_bytes, 0x00020000);

    // DMA map: 512 16-byte chunks per tile and operand, 2 per thread; source chunk swizzled, LDS image lane-linear
    const int srow = tid >> 3, spos = tid & 7;
    unsigned koff[2], voff[2];
    int vkey[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = srow + 32 * i;
        const int csrc = spos ^ ((r >> 1) & 7);
        koff[i] = (unsigned)((int64_t)r * ldk_s * 2) + csrc * 16;       // + kt*64 rows per tile
        voff[i] = (unsigned)((int64_t)r * ldvt_s * 2) + csrc * 16;      // + kt*128 bytes per tile
        vkey[i] = csrc * 8;                                             // first key of this V^T chunk inside the tile
    }
    const unsigned krow_bytes = (unsigned)(ldk_s * 2);
    auto load_tile = [&](int kt, int buf) {
        half_t* dk = &sK[buf][wave * 8 * 64];
        half_t* dv = &sV[buf][wave * 8 * 64];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kt * FK + srow + 32 * i;
            const unsigned kv = key < nk_s ? koff[i] + (unsigned)(kt * FK) * krow_bytes : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(dk + 32 * i * 64), 16, kv, 0, 0, 0);
            const unsigned vv = (kt * FK + vkey[i] < nk_s) ? voff[i] + (unsigned)(kt * FK * 2) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(dv + 32 * i * 64), 16, vv, 0, 0, 0);
        }
    };

    f16v oacc[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        m_run[b] = PRE ? 0.f : -1e30f;     // PRE: the first tile always moves the max (see below), 0 keeps its scores exact
        l_run[b] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[b][0][i] = 0.f; oacc[b][1][i] = 0.f; }
    }

    f16v cinit[PRE ? QB : 1];              // PRE: -m_run broadcast, the C operand of the first score MFMA
#pragma unroll
    for (int b = 0; b < (PRE ? QB : 1); ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) cinit[b][i] = 0.f;

    const int ntiles = (nk_s + FK - 1) / FK;
    load_tile(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) load_tile(kt + 1, cur ^ 1);
        const half_t* cK = sK[cur];
        const half_t* cV = sV[cur];

        // ---- S^T = K Q^T for the two 32-key halves of the tile; each K fragment feeds QB MFMAs.  The first MFMA of an
        // accumulator takes a literal zero C operand (no per-tile v_mov zeroing of 64 registers).
        f16v sacc[QB][2];
        const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const h8 kf = *reinterpret_cast<const h8*>(cK + tile_off(kb * 32 + lq, s * 2 + hi));
#pragma unroll
                for (int b = 0; b < QB; ++b)
                    sacc[b][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[b][s], s == 0 ? (PRE ? cinit[b] : zero16) : sacc[b][kb], 0, 0, 0);
            }
        // ---- online softmax per query block in the log2 domain: p = 2^(s*c - m), one fma + one v_exp per score.
        // Key masking exists only in the code path of a partial last tile (block-uniform branch).
        const int key_base = kt * FK;
        if (key_base + FK > nk_s) {
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key_base + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key >= nk_s) sacc[b][kb][r] = -1e30f;
                    }
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float mx = -1e30f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // Deferred max update: the running max (log2 units) moves only when some query of the wave would exceed it by more
            // than FLASH_DEFER; otherwise the tile is exponentiated against the old max (P <= 2^FLASH_DEFER: exact in the
            // fp32 sums, same relative precision in fp16) and the O / l rescale - 64 multiplies and an exponential per lane -
            // is skipped.  The decision precedes the exponentiation of the tile it covers and the previous tile's P V is
            // complete, so everything still at the old scale (O, l) is rescaled exactly once.
            float psum = 0.f;
            if (PRE) {
                // the scores are already s - m_run: the max moves by max(mx, 0) (by mx on the first tile, where nothing is
                // accumulated yet and m_run = 0 is only a placeholder)
                const bool first = kt == 0;
                if (first || __builtin_amdgcn_ballot_w64(mx > FLASH_DEFER) != 0) {      // wave-uniform branch
                    const float delta = first ? mx : fmaxf(mx, 0.f);
                    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
                    m_run[b] += delta;
                    l_run[b] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { oacc[b][0][i] *= alpha; oacc[b][1][i] *= alpha; }
#pragma unroll
                    for (int i = 0; i < 16; ++i) { sacc[b][0][i] -= delta; sacc[b][1][i] -= delta; cinit[b][i] = -m_run[b]; }
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(sacc[b][kb][r]);
                        sacc[b][kb][r] = pv;
                        psum += pv;
                    }
            } else {
            const float cand = mx * p.scale_log2;
            if (__builtin_amdgcn_ballot_w64(cand > m_run[b] + FLASH_DEFER) != 0) {     // wave-uniform branch
                const float m_new = fmaxf(m_run[b], cand);
                const float alpha = __builtin_amdgcn_exp2f(m_run[b] - m_new);
                m_run[b] = m_new;
                l_run[b] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[b][0][i] *= alpha; oacc[b][1][i] *= alpha; }
            }
            const float m_use = m_run[b];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][kb][r], p.scale_log2, -m_use));
                    sacc[b][kb][r] = pv;
                    psum += pv;
                }
            }
            l_run[b] += psum;
        }
        // ---- O^T += V^T P^T; each V^T fragment feeds QB MFMAs
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 pf[QB];
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[b][j] = (half_t)sacc[b][kb][8 * s + j];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int row = db * 32 + lq;
                    const int c0 = kb * 4 + 2 * s;
                    const h4 lo = *reinterpret_cast<const h4*>(cV + tile_off(row, c0) + 4 * hi);
                    const h4 hi4 = *reinterpret_cast<const h4*>(cV + tile_off(row, c0 + 1) + 4 * hi);
                    const h8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
                    for (int b = 0; b < QB; ++b) oacc[b][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[b], oacc[b][db], 0, 0, 0);
                }
            }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32);
        const float inv = 1.0f / l_tot;
        if (DUAL && set == 0) {          // keep softmax(Q K1^T) V1, go round again for the second set
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (QB == 1) keep[QB == 1 ? b : 0][db][i] = oacc[b][db][i] * inv;
                    else keep16[QB == 2 ? b : 0][db][i] = (half_t)(oacc[b][db][i] * inv);
                }
            continue;
        }
        if (qvalid[b]) {
            half_t* orow = p.o + ((int64_t)g * p.nq + q0 + b * 32 + lq) * p.ldo + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int d0 = db * 32 + 8 * gq + 4 * hi;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = oacc[b][db][gq * 4 + r] * inv;
                    if (DUAL) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += QB == 1 ? keep[QB == 1 ? b : 0][db][gq * 4 + r] : (float)keep16[QB == 2 ? b : 0][db][gq * 4 + r];
                    }
                    if (p.accumulate & VCX_ATTN_ACCUMULATE) {
                        const h4 old = *reinterpret_cast<const h4*>(orow + d0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
                    }
                    *reinterpret_cast<h4*>(orow + d0) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                }
        }
    }
  }
#endif
}

// =======================================================================================
// Text (+) image cross-attention with the keys and values RESIDENT in LDS.
// Both key sets of a (frame group, head) unit - 77 text keys and 256 image tokens shared by all frames of a video - are
// 6 tiles of 64 keys = 96 KB of K and V^T: a block of 8 waves loads them once and then streams over its share of the
// unit's queries (all frames x pixels), every wave on its own, with no barrier and no DMA in the loop.  The per-tile DMA +
// barrier pipeline of flash_d64_kernel costs more than the arithmetic when a query block only meets six key tiles.
// =======================================================================================
struct XAttnArgs {
    const half_t* q;
    half_t* o;
    const half_t* k1;
    const half_t* vt1;
    const half_t* k2;
    const half_t* vt2;
    int heads, nk1, nk2, kv_rows1, kv_rows2, split, nunits;
    int64_t ldq, ldo, ldk1, ldvt1, ldk2, ldvt2;
    int64_t rows_per_unit, rows_per_block;      // queries of one (group, head) unit = frames * nq; a block's share (x 512)
    float scale_log2;
};

__global__ void __launch_bounds__(512, 1) xattn_resident_d64_kernel(XAttnArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int T1 = 2, T2 = 4, NT = T1 + T2;              // key tiles: text (<= 128 keys), image (<= 256 keys)
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];
    half_t* sK = reinterpret_cast<half_t*>(xsmem);           // [NT][64 keys][64] swizzled
    half_t* sV = sK + NT * 64 * 64;                          // [NT][64 d][64 keys] swizzled

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    const int unit = blockIdx.x / p.split, part = blockIdx.x % p.split;
    const int gb = unit / p.heads, h = unit % p.heads;

    // ---- K / V^T of both sets -> LDS (8 rows per DMA instruction, 16 instructions per tile, spread over the 8 waves)
    {
        const int srow8 = lane >> 3, spos = lane & 7;
        for (int idx = wave; idx < NT * 16; idx += 8) {
            const int t = idx >> 4, j = idx & 15;
            const bool second = t >= T1;
            const int kt = second ? t - T1 : t;
            const int nk = second ? p.nk2 : p.nk1;
            const int64_t ldk = second ? p.ldk2 : p.ldk1, ldvt = second ? p.ldvt2 : p.ldvt1;
            const int64_t kvrow0 = (int64_t)gb * (second ? p.kv_rows2 : p.kv_rows1);
            const half_t* kbase = (second ? p.k2 : p.k1) + kvrow0 * ldk + h * 64;
            const half_t* vbase = (second ? p.vt2 : p.vt1) + (int64_t)(h * 64) * ldvt + kvrow0;
            const int r = (j & 7) * 8 + srow8;                 // row inside the tile: key (K part) or d (V^T part)
            const int csrc = spos ^ ((r >> 1) & 7);
            if (j < 8) {
                const unsigned bytes = (unsigned)(((int64_t)(nk - 1) * ldk + 64) * 2);
                const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kbase), 0, (int)bytes, 0x00020000);
                const int key = kt * 64 + r;
                const unsigned v = key < nk ? (unsigned)((int64_t)key * ldk * 2) + csrc * 16 : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(sK + t * 4096 + (j & 7) * 512), 16, v, 0, 0, 0);
            } else {
                const unsigned bytes = (unsigned)((63ll * ldvt + ((nk + 7) & ~7)) * 2);
                const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vbase), 0, (int)bytes, 0x00020000);
                const int key0 = kt * 64 + csrc * 8;
                const unsigned v = key0 < nk ? (unsigned)((int64_t)r * ldvt * 2) + (unsigned)(kt * 128) + csrc * 16 : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(sV + t * 4096 + (j & 7) * 512), 16, v, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }

    const int64_t row_begin = (int64_t)part * p.rows_per_block;
    const int64_t row_end = row_begin + p.rows_per_block < p.rows_per_unit ? row_begin + p.rows_per_block : p.rows_per_unit;
    const half_t* qbase = p.q + ((int64_t)gb * p.rows_per_unit) * p.ldq + h * 64;
    half_t* obase = p.o + ((int64_t)gb * p.rows_per_unit) * p.ldo + h * 64;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typedef half_t h16v __attribute__((ext_vector_type(16)));

    for (int64_t r0 = row_begin + wave * 64; r0 < row_end; r0 += 512) {       // this wave's two 32-row query blocks
        h8 qf[2][4];
        bool qvalid[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t qrow = r0 + b * 32 + lq;
            qvalid[b] = qrow < row_end;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                qf[b][s] = qvalid[b] ? *reinterpret_cast<const h8*>(qbase + qrow * p.ldq + s * 16 + hi * 8) : zero8;
        }
        h16v keep[2][2];
#pragma unroll 1
        for (int set = 0; set < 2; ++set) {
            const int t0 = set ? T1 : 0, nt = set ? T2 : T1, nk = set ? p.nk2 : p.nk1;
            f16v oacc[2][2];
            float m_run[2], l_run[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                m_run[b] = -1e30f;
                l_run[b] = 0.f;
                oacc[b][0] = zero16;
                oacc[b][1] = zero16;
            }
#pragma unroll 1
            for (int kt = 0; kt < nt; ++kt) {
                if (kt * 64 >= nk) break;
                const half_t* cK = sK + (t0 + kt) * 4096;
                const half_t* cV = sV + (t0 + kt) * 4096;
                f16v sacc[2][2];
                const int key_base = kt * 64;
                const bool half2 = key_base + 32 < nk;           // wave-uniform: the tile's second 32 keys hold a valid key
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (kb == 1 && !half2) {                     // e.g. 77 text keys: keys 96..127 do not exist
                        sacc[0][1] = zero16;
                        sacc[1][1] = zero16;
                        break;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const h8 kf = *reinterpret_cast<const h8*>(cK + tile_off(kb * 32 + lq, s * 2 + hi));
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            sacc[b][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[b][s], s == 0 ? zero16 : sacc[b][kb], 0, 0, 0);
                    }
                }
                if (key_base + 64 > nk) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int key = key_base + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                if (key >= nk) sacc[b][kb][r] = -1e30f;
                            }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float mx = -1e30f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][kb][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float cand = mx * p.scale_log2;
                    if (__builtin_amdgcn_ballot_w64(cand > m_run[b] + FLASH_DEFER) != 0) {     // deferred max, as in flash_d64_kernel
                        const float m_new = fmaxf(m_run[b], cand);
                        const float alpha = __builtin_amdgcn_exp2f(m_run[b] - m_new);
                        m_run[b] = m_new;
                        l_run[b] *= alpha;
#pragma unroll
                        for (int i = 0; i < 16; ++i) { oacc[b][0][i] *= alpha; oacc[b][1][i] *= alpha; }
                    }
                    const float m_use = m_run[b];
                    float psum = 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][kb][r], p.scale_log2, -m_use));
                            sacc[b][kb][r] = pv;
                            psum += pv;
                        }
                    l_run[b] += psum;
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (kb == 1 && !half2) break;                // all-zero probabilities: nothing to add
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        h8 pf[2];
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int j = 0; j < 8; ++j) pf[b][j] = (half_t)sacc[b][kb][8 * s + j];
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const int row = db * 32 + lq;
                            const int c0 = kb * 4 + 2 * s;
                            const h4 lo = *reinterpret_cast<const h4*>(cV + tile_off(row, c0) + 4 * hi);
                            const h4 hi4 = *reinterpret_cast<const h4*>(cV + tile_off(row, c0 + 1) + 4 * hi);
                            const h8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
                            for (int b = 0; b < 2; ++b) oacc[b][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[b], oacc[b][db], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- normalise; keep the first set's result, add and store after the second
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32);
                const float inv = 1.0f / l_tot;
                if (set == 0) {
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int i = 0; i < 16; ++i) keep[b][db][i] = (half_t)(oacc[b][db][i] * inv);
                } else if (qvalid[b]) {
                    half_t* orow = obase + (r0 + b * 32 + lq) * p.ldo;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int d0 = db * 32 + 8 * gq + 4 * hi;
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = oacc[b][db][gq * 4 + r] * inv + (float)keep[b][db][gq * 4 + r];
                            *reinterpret_cast<h4*>(orow + d0) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        }
                }
            }
        }
    }
#endif
}

// =======================================================================================
// Resident cross-attention, second form (round 5).  Same residency, same arithmetic per score as xattn_resident_d64_kernel; what
// changes is everything around the MFMAs, which ran at 0.2 of either roof (0.338 ms at level 0 for 116 us of matrix work):
//   * the loop walks HALF tiles (32 keys): one score accumulator per query block instead of two, and the 32 registers that frees
//     hold the K fragments of the NEXT half tile and the V^T fragments of this one, requested a whole MFMA / softmax phase ahead
//     (v1: ds_read -> lgkmcnt(0) -> two MFMAs, eight times per tile);
//   * the Q rows of the wave's NEXT 64 queries are requested (branch-free buffer loads, rows past the end arrive as zeros without
//     traffic) when an iteration starts and used when the next one does - v1 waited vmcnt(0) for them at the top of every iteration;
//   * the first key set's normalised result waits in a wave-private LDS patch (8 KB x 8 waves = the 64 KB the K / V^T image leaves
//     free) instead of 32 registers;
//   * the deferred-max test is a ballot over the per-lane PARTIAL maxima (a row exceeds the threshold iff one of its two lanes
//     does); the exact row maximum - v_permlane32_swap, not an LDS round trip - is only formed inside the rare rescale;
//   * output as 16-byte buffer stores (lane pairs exchange halves with v_permlane32_swap), rows past the end dropped by the range
//     check: no exec-mask branches, so hipcc's vector-memory counts stay exact and the Q wait does not drain the stores.
// A text half tile that holds no key (77 keys: 3 of 4) is never visited, so nothing is special-cased for it.
// =======================================================================================
// Both halves of a query row: lanes l and l ^ 32 each hold a partial value x; returns (x of lanes 0-31, x of lanes 32-63) in every
// lane.  The two results are copied into scalars BEFORE the bit cast: __builtin_bit_cast(float, sw[1]) on the element of the
// builtin's vector result reads element 0 with this hipcc (the IR holds a single extractvalue 0; seen in the listing as
// "rcp(x) * 0.5" for 1 / (lo + hi)) - the first GPU run of this kernel normalised every row with twice one half's sum.
__device__ __forceinline__ void row_halves(float x, float& lo, float& hi) {
    const unsigned a = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned r0 = sw[0], r1 = sw[1];
    lo = __builtin_bit_cast(float, r0);
    hi = __builtin_bit_cast(float, r1);
}

// (XABL: timing-only ablation bits of this kernel, 0 in the product - vcx_ablate.h)
__global__ void __launch_bounds__(512, 1) xattn_resident2_d64_kernel(XAttnArgs p, unsigned q_bytes, unsigned o_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int T1 = 2, T2 = 4, NT = T1 + T2;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];
    half_t* sK = reinterpret_cast<half_t*>(xsmem);           // [NT][64 keys][64] swizzled
    half_t* sV = sK + NT * 64 * 64;                          // [NT][64 d][64 keys] swizzled

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    const int unit = blockIdx.x / p.split, part = blockIdx.x % p.split;
    const int gb = unit / p.heads, h = unit % p.heads;
    u4v* sKeep = reinterpret_cast<u4v*>(xsmem + 2 * NT * 8192 + wave * 8192) + lane;      // piece i of this lane: sKeep[64 i], i < 8

    // ---- K / V^T of both sets -> LDS (as in xattn_resident_d64_kernel)
    {
        const int srow8 = lane >> 3, spos = lane & 7;
        for (int idx = wave; idx < NT * 16; idx += 8) {
            const int t = idx >> 4, j = idx & 15;
            const bool second = t >= T1;
            const int kt = second ? t - T1 : t;
            const int nk = second ? p.nk2 : p.nk1;
            const int64_t ldk = second ? p.ldk2 : p.ldk1, ldvt = second ? p.ldvt2 : p.ldvt1;
            const int64_t kvrow0 = (int64_t)gb * (second ? p.kv_rows2 : p.kv_rows1);
            const half_t* kbase = (second ? p.k2 : p.k1) + kvrow0 * ldk + h * 64;
            const half_t* vbase = (second ? p.vt2 : p.vt1) + (int64_t)(h * 64) * ldvt + kvrow0;
            const int r = (j & 7) * 8 + srow8;
            const int csrc = spos ^ ((r >> 1) & 7);
            if (j < 8) {
                const unsigned bytes = (unsigned)(((int64_t)(nk - 1) * ldk + 64) * 2);
                const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kbase), 0, (int)bytes, 0x00020000);
                const int key = kt * 64 + r;
                const unsigned v = key < nk ? (unsigned)((int64_t)key * ldk * 2) + csrc * 16 : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(sK + t * 4096 + (j & 7) * 512), 16, v, 0, 0, 0);
            } else {
                const unsigned bytes = (unsigned)((63ll * ldvt + ((nk + 7) & ~7)) * 2);
                const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vbase), 0, (int)bytes, 0x00020000);
                const int key0 = kt * 64 + csrc * 8;
                const unsigned v = key0 < nk ? (unsigned)((int64_t)r * ldvt * 2) + (unsigned)(kt * 128) + csrc * 16 : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(sV + t * 4096 + (j & 7) * 512), 16, v, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }

    const int64_t row_begin = (int64_t)part * p.rows_per_block;
    const int64_t row_end = row_begin + p.rows_per_block < p.rows_per_unit ? row_begin + p.rows_per_block : p.rows_per_unit;
    const half_t* qbase = p.q + ((int64_t)gb * p.rows_per_unit) * p.ldq + h * 64;
    half_t* obase = p.o + ((int64_t)gb * p.rows_per_unit) * p.ldo + h * 64;
    const __amdgpu_buffer_rsrc_t srd_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(qbase), 0, (int)q_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)o_bytes, 0x00020000);
    const unsigned ldq2 = (unsigned)p.ldq * 2u, ldo2 = (unsigned)p.ldo * 2u;
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float scale = p.scale_log2;
    const int nh1 = (p.nk1 + 31) >> 5, nh = nh1 + ((p.nk2 + 31) >> 5);          // half tiles of the first set, of both

    // fragment offsets (elements) inside a tile for key half 0: K fragment of k-step s = row lq, chunk 2 s + hi; V^T fragment
    // piece (s, e) = row lq (+ 32 db), chunk 2 s + e, halves 4 hi .. + 3.  Key half 1: K 32 rows on, V^T chunk + 4 = offset ^ 32.
    int ko[4], vo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        ko[s] = tile_off(lq, 2 * s + hi);
        vo[s] = tile_off(lq, s) + 4 * hi;
    }
    auto read_k = [&](int j, h8 (&kf)[4]) {                 // half tile j of the flat sequence (both sets)
        const int jj = j >= nh1 ? j - nh1 : j;
        const half_t* base = sK + ((j >= nh1 ? T1 : 0) + (jj >> 1)) * 4096 + (jj & 1) * 2048;
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const h8*>(base + ko[s]);
    };
    auto load_q = [&](int64_t r0, u4v (&dst)[8]) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t qrow = r0 + b * 32 + lq;
            const unsigned v = qrow < row_end ? (unsigned)qrow * ldq2 + (unsigned)hi * 16u : OOB;
#pragma unroll
            for (int s = 0; s < 4; ++s) dst[b * 4 + s] = __builtin_amdgcn_raw_buffer_load_b128(srd_q, v, s * 32, 0);
        }
    };

    u4v qn[8];
    h8 kf[4];
    load_q(row_begin + wave * 64, qn);
    read_k(0, kf);
    for (int64_t r0 = row_begin + wave * 64; r0 < row_end; r0 += 512) {       // this wave's two 32-row query blocks
        h8 qf[2][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i >> 2][i & 3] = __builtin_bit_cast(h8, qn[i]);
#if !(XABL & 8)
        load_q(r0 + 512, qn);                                                 // the next iteration's rows (zeros past the end)
#endif
        f16v oacc[2][2];
        float m_run[2], l_run[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            m_run[b] = -1e30f;
            l_run[b] = 0.f;
            oacc[b][0] = zero16;
            oacc[b][1] = zero16;
        }
#pragma unroll 1
        for (int j = 0; j < nh; ++j) {
            if (j == nh1) {
                // ---- first set complete: normalise, park in LDS as fp16, start over for the second set
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float l_lo, l_hi;
                    row_halves(l_run[b], l_lo, l_hi);
                    const float inv = 1.0f / (l_lo + l_hi);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        h8 lo8, hi8;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            lo8[i] = (half_t)(oacc[b][db][i] * inv);
                            hi8[i] = (half_t)(oacc[b][db][8 + i] * inv);
                        }
                        const u4v w0 = __builtin_bit_cast(u4v, lo8), w1 = __builtin_bit_cast(u4v, hi8);
                        sKeep[64 * (4 * b + 2 * db)] = w0;
                        sKeep[64 * (4 * b + 2 * db + 1)] = w1;
                        // a wide store still reads its data registers one or two slots on (tools/isa_audit.py, the library-wide rule of
                        // round 3): the data stay live, and untouched, across two wait states
                        asm volatile("s_nop 1" : : "v"(w0), "v"(w1));
                        oacc[b][db] = zero16;
                    }
                    m_run[b] = -1e30f;
                    l_run[b] = 0.f;
                }
            }
            const bool second = j >= nh1;
            const int jj = second ? j - nh1 : j;
            const int kb = jj & 1, key_base = jj * 32, nk = second ? p.nk2 : p.nk1;
            const half_t* cV = sV + ((second ? T1 : 0) + (jj >> 1)) * 4096;
            // ---- V^T fragments of this half tile (used behind the softmax), then the scores
            h8 vf[2][2];
#if XABL & 16
#pragma unroll
            for (int s = 0; s < 2; ++s) { vf[s][0] = qf[0][s]; vf[s][1] = qf[1][s]; }
            if (0)
#endif
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const h4 lo = *reinterpret_cast<const h4*>(cV + db * 2048 + (vo[2 * s] ^ (kb * 32)));
                    const h4 hi4 = *reinterpret_cast<const h4*>(cV + db * 2048 + (vo[2 * s + 1] ^ (kb * 32)));
                    vf[s][db] = h8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                }
            f16v sacc[2];
#if XABL & 4
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[b][r] = (float)kf[r & 3][r >> 2] + (float)qf[b][r & 3][r >> 2];
#else
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    sacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qf[b][s], s == 0 ? zero16 : sacc[b], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            // ---- K fragments of the next half tile (of the next iteration's first one behind the last): a softmax ahead of their use
#if !(XABL & 16)
            read_k(j + 1 < nh ? j + 1 : 0, kf);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (key_base + 32 > nk) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key >= nk) sacc[b][r] = -1e30f;
                    }
            }
#if XABL & 2
            if (0)
#endif
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float mx = sacc[b][0];
#if XABL & 32
                mx = 0.f;
                if (0)
#endif
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[b][r]);
                const float cand = mx * scale;                                           // this lane's 16 of the row's 32 keys
                if (!(XABL & 32) && __builtin_amdgcn_ballot_w64(cand > m_run[b] + FLASH_DEFER) != 0) {   // deferred max, as in flash_d64_kernel
                    float c_lo, c_hi;
                    row_halves(cand, c_lo, c_hi);
                    const float m_new = fmaxf(m_run[b], fmaxf(c_lo, c_hi));
                    const float alpha = __builtin_amdgcn_exp2f(m_run[b] - m_new);
                    m_run[b] = m_new;
                    l_run[b] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { oacc[b][0][i] *= alpha; oacc[b][1][i] *= alpha; }
                }
                const float m_use = m_run[b];
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#if XABL & 1
                    const float pv = __builtin_fmaf(sacc[b][r], scale, -m_use);
#else
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][r], scale, -m_use));
#endif
                    sacc[b][r] = pv;
                    psum += pv;
                }
                l_run[b] += psum;
            }
            // ---- O^T += V^T P^T
#if XABL & 4
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[b][0][r] += sacc[b][r] * (float)vf[0][0][r & 7]; oacc[b][1][r] += sacc[b][r] * (float)vf[1][1][r & 7]; }
            if (0)
#endif
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 pf[2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 8; ++i) pf[b][i] = (half_t)sacc[b][8 * s + i];
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int b = 0; b < 2; ++b) oacc[b][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s][db], pf[b], oacc[b][db], 0, 0, 0);
            }
        }
        // ---- second set complete: normalise, add the first set's result, store
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float l_lo, l_hi;
            row_halves(l_run[b], l_lo, l_hi);
            const float inv = 1.0f / (l_lo + l_hi);
            const int64_t qrow = r0 + b * 32 + lq;
            const unsigned ov = qrow < row_end ? (unsigned)qrow * ldo2 + (unsigned)hi * 16u : OOB;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const h8 k0 = __builtin_bit_cast(h8, sKeep[64 * (4 * b + 2 * db)]);
                const h8 k1 = __builtin_bit_cast(h8, sKeep[64 * (4 * b + 2 * db + 1)]);
                u2v pk[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    h4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = gq * 4 + r;
                        o4[r] = (half_t)(oacc[b][db][i] * inv + (float)(i < 8 ? k0[i] : k1[i - 8]));
                    }
                    pk[gq] = __builtin_bit_cast(u2v, o4);
                }
                // column group k = 4 db + gq holds columns 8 k + 4 hi .. + 3 of this lane's row: after the half swap lanes 0-31 own
                // columns 8 k .. 8 k + 7 and lanes 32-63 columns 8 k + 8 .. 8 k + 15 (as tattn_d64_kernel)
#pragma unroll
                for (int gq = 0; gq < 4; gq += 2) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[gq][0], pk[gq + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[gq][1], pk[gq + 1][1], false, false);
                    const u4v w = {s0[0], s1[0], s0[1], s1[1]};
#if XABL & 8
                    if (w[0] == 0x12345678u)
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(w, srd_o, ov, (db * 32 + gq * 8) * 2, 0);
                    asm volatile("s_nop 1" : : "v"(w));          // (the wide-store rule again)
                }
            }
        }
    }
#endif
}

// =======================================================================================
// Flash attention, ONE head of d = 512: the AttnBlock of the VAE (reference lvdm/modules/networks/ae_modules.py:26-78: 9216 tokens
// per 576x1024 frame, softmax(q k^T / sqrt(512)) v).  The score matrix is never written: 170 MB per frame as fp16, which the
// GEMM -> row softmax -> GEMM sequence of rounds 1-4 wrote, read, rewrote and read again, frame by frame.
//
// Swapped formulation as in flash_d64_kernel (S^T = K Q^T, O^T += V^T P^T: the softmax statistics of a query live in the four lanes
// q, q + 16, q + 32, q + 48 and P feeds the second product without cross-lane movement), on v_mfma_f32_16x16x32_f16 so that a wave's
// state fits 256 registers: a wave owns 16 queries with Q (16 x 512 fp16 = 64 registers) and O^T (512 x 16 fp32 = 128 registers)
// resident.  Eight waves per block (128 queries), two per SIMD; K / V^T tiles of 32 keys go HBM -> LDS by DMA, two stages of 64 KB:
// K as eight [32 keys][64 d] slabs, V^T as eight [64 d][32 keys] slabs with two d rows per 128-byte line, both in the XOR-swizzled
// line layout of tile_off().  Per tile a wave issues 32 score MFMAs and 32 PV MFMAs around ~70 VALU instructions of softmax: the
// contraction is eight times as long as at d = 64, the phased order costs little, and the kernel is plain HIP.  LDS traffic is the
// bound (1 KB fragment per MFMA).
// =======================================================================================
struct Flash512Args {
    const half_t* q;
    const half_t* k;
    const half_t* vt;
    half_t* o;
    int nq, nk, kv_rows, nqb, nprob;
    int64_t ldq, ldk, ldvt, ldo;
    float scale_log2;
};

// the four lanes q + 16 g of a 16-lane row: all-reduce by two butterfly steps (DPP row_shr is confined to 16 lanes: permute through LDS hardware)
__device__ __forceinline__ float quad_rows_max(float x) {
    x = fmaxf(x, __shfl_xor(x, 16));
    return fmaxf(x, __shfl_xor(x, 32));
}
__device__ __forceinline__ float quad_rows_sum(float x) {
    x += __shfl_xor(x, 16);
    return x + __shfl_xor(x, 32);
}

__global__ void __launch_bounds__(512, 1) flash_d512_kernel(Flash512Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int D = 512, TK = 32, STAGE = 2 * TK * D;           // elements per stage: K tile + V^T tile
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char fsmem[];
    half_t* smem = reinterpret_cast<half_t*>(fsmem);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lr = lane & 15, lg = lane >> 4;
    // all query blocks of one frame on ONE XCD (ids of the same residue mod 8): its K / V^T stream is shared through that L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = (slot / p.nqb) * 8 + xcd;
    if (g >= p.nprob) return;
    const int q0 = ((slot % p.nqb) * 8 + wave) * 16;

    // ---- Q fragments (B operand of the score MFMAs): lane (q = lr, lg) holds Q[q][32 s + 8 lg .. + 7], s = 0..15
    const half_t* qbase = p.q + ((int64_t)g * p.nq) * p.ldq;
    const int qrow = q0 + lr;
    const bool qvalid = qrow < p.nq;
    const half_t* qptr = qbase + (int64_t)(qvalid ? qrow : p.nq - 1) * p.ldq + lg * 8;
    h8 qf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) qf[s] = *reinterpret_cast<const h8*>(qptr + s * 32);

    // ---- K / V^T streams
    const half_t* kbase = p.k + ((int64_t)g * p.kv_rows) * p.ldk;
    const half_t* vbase = p.vt + (int64_t)g * p.kv_rows;
    const unsigned k_bytes = (unsigned)(((int64_t)(p.nk - 1) * p.ldk + D) * 2);
    const unsigned v_bytes = (unsigned)(((int64_t)(D - 1) * p.ldvt + ((p.nk + 7) & ~7)) * 2);
    const __amdgpu_buffer_rsrc_t srd_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kbase), 0, (int)k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vbase), 0, (int)v_bytes, 0x00020000);
    // DMA map: a slab is 32 lines of 128 bytes = 256 chunks of 16 bytes; thread t takes chunk t & 255 of slabs (t >> 8) + 2 c, c < 4.
    // LDS line L = (t & 255) >> 3, position t & 7 (lane-linear image), source chunk j = position ^ ((L >> 1) & 7).
    // K slab: line = key, j = d / 8 inside the slab.  V^T slab: line = two d rows, j = 4 (d & 1) + keys / 8.
    const int dt = tid & 255, dh = tid >> 8;
    const int dl = dt >> 3, dc = (dt & 7) ^ ((dl >> 1) & 7);
    const unsigned koff = (unsigned)((int64_t)dl * p.ldk * 2) + dc * 16 + dh * 128;                                          // + tile * 32 rows; + 256 bytes per c
    const unsigned voff = (unsigned)((int64_t)(2 * dl + (dc >> 2) + 64 * dh) * p.ldvt * 2) + (dc & 3) * 16;                  // + tile * 64 bytes; + 128 d rows per c
    const unsigned ktile_bytes = (unsigned)(TK * p.ldk * 2), vslab2_bytes = (unsigned)(128 * p.ldvt * 2);
    const int wl = (dt >> 6) * 512;                                 // this wave's 8 lines inside a slab (elements)
    auto load_tile = [&](int kt, int buf) {
        half_t* dk = smem + buf * STAGE + dh * 2048 + wl;
        half_t* dv = dk + TK * D;
        const bool kin = kt * TK + dl < p.nk;                       // key row exists
        const bool vin = kt * TK + (dc & 3) * 8 < p.nk;             // first key of the V^T chunk exists (nk rounded up to 8 in the extent)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(dk + c * 4096), 16, kin ? koff + (unsigned)kt * ktile_bytes : 0xFFFFFFFFu, c * 256, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(dv + c * 4096), 16, vin ? voff + (unsigned)kt * 64u + (unsigned)c * vslab2_bytes : 0xFFFFFFFFu, 0, 0, 0);
        }
    };

    // fragment offsets (elements) inside a slab.  K (A operand, 16 keys x 32 d): line = key 16 kb + lr, chunk 4 (s & 1) + lg of slab
    // s >> 1.  V^T (A operand, 16 d rows x 32 keys): d row 16 i + lr inside the slab (i < 4): line 8 i + (lr >> 1), chunk
    // 4 (lr & 1) + c; this lane's eight keys are 4 lg .. + 3 (chunk lg >> 1, halves 4 (lg & 1) ..) and 16 + 4 lg .. + 3 (chunk 2 + (lg >> 1)).
    int ko[2], vo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ko[i] = tile_off(lr, 4 * i + lg);
        vo[i] = tile_off(lr >> 1, 4 * (lr & 1) + 2 * i + (lg >> 1)) + 4 * (lg & 1);
    }

    f4 oacc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) oacc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;
    const float scale = p.scale_log2;

    const int ntiles = (p.nk + TK - 1) / TK;
    load_tile(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) load_tile(kt + 1, cur ^ 1);
        const half_t* cK = smem + cur * STAGE;
        const half_t* cV = cK + TK * D;
        // ---- S^T = K Q^T: two blocks of 16 keys, 16 k-steps of 32 d each
        f4 sacc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const h8 kf = *reinterpret_cast<const h8*>(cK + (s >> 1) * 2048 + kb * 1024 + ko[s & 1]);
                sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], sacc[kb], 0, 0, 0);
            }
        // lane (q = lr, lg) holds the scores of keys 16 kb + 4 lg + r
        const int key_base = kt * TK;
        if (key_base + TK > p.nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key_base + 16 * kb + 4 * lg + r >= p.nk) sacc[kb][r] = -1e30f;
        }
        // ---- online softmax (base 2, deferred running max as in flash_d64_kernel: the test is a ballot over the per-lane partial maxima)
        float mx = fmaxf(fmaxf(fmaxf(sacc[0][0], sacc[0][1]), fmaxf(sacc[0][2], sacc[0][3])), fmaxf(fmaxf(sacc[1][0], sacc[1][1]), fmaxf(sacc[1][2], sacc[1][3])));
        const float cand = mx * scale;
        if (__builtin_amdgcn_ballot_w64(cand > m_run + FLASH_DEFER) != 0) {
            const float m_new = fmaxf(m_run, quad_rows_max(cand));
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 32; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[i][r] *= alpha;
        }
        h8 pf;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][r], scale, -m_run));
                psum += pv;
                pf[4 * kb + r] = (half_t)pv;
            }
        l_run += psum;
        // ---- O^T += V^T P^T: 32 blocks of 16 d rows, one k-step of 32 keys (this lane's k-slots: keys 4 lg .. + 3, 16 + 4 lg .. + 3)
#pragma unroll
        for (int db = 0; db < 32; ++db) {
            // slab db / 4, d rows 16 (db & 3) + lr: 8 (db & 3) lines on - for an odd db & 3 that adds 4 to the line's swizzle term
            // ((line >> 1) & 7, < 4 for the first eight lines): chunk ^ 4 = element offset ^ 32
            const half_t* vrow = cV + (db >> 2) * 2048 + (db & 3) * 512;
            const h4 lo = *reinterpret_cast<const h4*>(vrow + (vo[0] ^ ((db & 1) * 32)));
            const h4 hi4 = *reinterpret_cast<const h4*>(vrow + (vo[1] ^ ((db & 1) * 32)));
            const h8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, oacc[db], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        cur ^= 1;
    }

    // ---- O[q][d] = O^T[d][q] / l: lane (q = lr, lg) holds d = 16 db + 4 lg + r: 8-byte stores
    const float inv = 1.0f / quad_rows_sum(l_run);
    if (qvalid) {
        half_t* orow = p.o + ((int64_t)g * p.nq + qrow) * p.ldo + lg * 4;
#pragma unroll
        for (int db = 0; db < 32; ++db)
            *reinterpret_cast<u2v*>(orow + db * 16) = __builtin_bit_cast(u2v, h4{(half_t)(oacc[db][0] * inv), (half_t)(oacc[db][1] * inv),
                                                                              (half_t)(oacc[db][2] * inv), (half_t)(oacc[db][3] * inv)});
    }
#endif
}

// =======================================================================================
// Temporal attention: T <= 32 frames, d = 64.
// =======================================================================================
struct TAttnArgs {
    const half_t* qkv;
    half_t* o;
    int B, T, heads;
    int64_t P, ld, ldo;
    int k_off, v_off;
    float scale;
    int64_t npairs;
    // relative position (tattn_d64_kernel<.., REL>; behind everything else, so that the other kernels' argument offsets stay):
    // relg [(b t p)][heads][64] = q . Ek[rel] per query, relp [(b t p)][heads][64] <- the probabilities by clipped distance, R = max distance
    const half_t* relg;
    half_t* relp;
    int R;
};

// One wave per (pixel, head): S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16_f16 (T <= 32 keys = one MFMA
// tile), the same swapped formulation as the flash kernel, so softmax statistics are per-lane and P needs no
// cross-lane movement.  Q, K and V rows (a frame row of one head is one 128-byte line) go through a wave-private LDS
// patch 3 x [32][72]: the Q / K fragments are ds_read_b128 from it, the V^T fragments are gathered with 16-bit reads
// (rows >= T zero-filled).  Traffic is the algorithmic minimum (q, k, v read once, o written once).
// CAUSAL (VCX_ATTN_CAUSAL; TemporalTransformer(causal_attention=True), reference attention.py:343-345, 377-384, 111-115 - not used by the ViewCrafter
// YAMLs): frame t attends to frames <= t.  An own instantiation: the unmasked kernel keeps its listing.
// REL (CrossAttention(relative_position=True), reference attention.py:20-40, 59-62, 104-108, 120-123 - `use_relative_position`, not used by the
// ViewCrafter YAMLs): logits += q_t . Ek[clamp(s - t, -R, R) + R] and out += sum_s P[t, s] Ev[clamp(s - t) + R].  Both tables have 2R + 1 <= 64 rows, so
// the two extra contractions are plain 64-wide GEMMs of the caller around this kernel: it ADDS the row relg[query][.] (= q Ek^T, one value per
// clipped distance) to its scores before scale and softmax, and WRITES the probabilities of a query by clipped distance (all keys beyond +-R summed
// into the end slots) to relp[query][.] - which the caller has zeroed and then multiplies with Ev.  Own instantiations.
template <bool CAUSAL, bool REL = false>
__global__ void __launch_bounds__(512) tattn_d64_kernel(TAttnArgs p) {
    constexpr int VLD = 72;                                           // LDS row pitch (halfs): 144 B keeps 16-B alignment
    constexpr int WAVES = 8;
    // wave-private patch: the V rows of this (pixel, head), 32 x 64 halfs, and ONE more such area that holds the Q rows and then -
    // once their fragments are in registers - the K rows (round 4: 9.2 KB per wave instead of 13.8, so that two blocks fit a CU
    // and sixteen waves, not eight, hide each other's HBM round trips)
    __shared__ __attribute__((aligned(16))) half_t sVt[WAVES][2 * 32 * VLD];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    const int64_t pair = (int64_t)blockIdx.x * WAVES + wave;
    if (pair >= p.npairs) return;                                     // wave-uniform; no block-level sync below
    const int h = (int)(pair % p.heads);
    const int64_t pix = (pair / p.heads) % p.P;
    const int b = (int)(pair / (p.heads * p.P));
    const half_t* base = p.qkv + ((int64_t)b * p.T * p.P + pix) * p.ld + h * 64;   // frame-0 row of this (pixel, head)
    const int64_t fstride = p.P * p.ld;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool rvalid = lq < p.T;

    // Q, K and V rows all arrive as whole 128-byte lines (8 lanes x 16 B per frame row) and go through the wave-private LDS
    // patch; the MFMA fragments are ds_read_b128 from there.  (Round 1 loaded the Q / K fragments straight from global memory -
    // 32 bytes from each of 32 frame rows per instruction: 4.2 TB/s; staged: 4.95 TB/s, profiles/r02_experiments.md.)
    h8 qf[4], kf[4];
    half_t* sv = sVt[wave];
    half_t* sqk = sv + 32 * VLD;
    {
        h8 q4[4], k4[4], v4[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = (lane >> 3) + 8 * it, ch = lane & 7;
            const half_t* src = base + (int64_t)row * fstride + ch * 8;
            const bool ok = row < p.T;
            q4[it] = ok ? *reinterpret_cast<const h8*>(src) : zero8;
            k4[it] = ok ? *reinterpret_cast<const h8*>(src + p.k_off) : zero8;
            v4[it] = ok ? *reinterpret_cast<const h8*>(src + p.v_off) : zero8;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = (lane >> 3) + 8 * it, ch = lane & 7;
            *reinterpret_cast<h8*>(sqk + row * VLD + ch * 8) = q4[it];
            *reinterpret_cast<h8*>(sv + row * VLD + ch * 8) = v4[it];
        }
        // LDS operations of one wave complete in order: no barrier (the patch is wave-private), and the K rows may overwrite the Q
        // rows as soon as the Q fragment reads have been ISSUED
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const h8*>(sqk + lq * VLD + s * 16 + hi * 8);
        asm volatile("" ::: "memory");                                // keep the compiler from moving the K stores above the Q reads
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = (lane >> 3) + 8 * it, ch = lane & 7;
            *reinterpret_cast<h8*>(sqk + row * VLD + ch * 8) = k4[it];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const h8*>(sqk + lq * VLD + s * 16 + hi * 8);
    }

    // S^T[key, q]: lane (q = lq, hi) holds keys (r&3) + 8*(r>>2) + 4*hi
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16v sacc = zero16;
#pragma unroll
    for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qf[s], s == 0 ? zero16 : sacc, 0, 0, 0);
    const float c = p.scale * 1.4426950408889634f;
    [[maybe_unused]] const int64_t relrow = REL ? ((((int64_t)b * p.T + (lq < p.T ? lq : p.T - 1)) * p.P + pix) * p.heads + h) * 64 : 0;
    if (REL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int dist = min(max(key - lq, -p.R), p.R) + p.R;
            if (key < p.T) sacc[r] += (float)p.relg[relrow + dist];
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= p.T || (CAUSAL && key > lq)) sacc[r] = -1e30f;
        mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * c;
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c, -mx));
        sacc[r] = e;
        l += e;
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    h8 pf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[s][j] = (half_t)(sacc[8 * s + j] * inv);   // normalised P, fp16 like the reference's einsum input
    if (REL) {
        // the probabilities of this lane's query by clipped distance: unique inside (-R, R), summed beyond (this lane's 16 keys, then the
        // other half of the row in lane ^ 32); masked / padded keys have probability 0
        float lo = 0.f, up = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int d = key - lq;
            const float pv = (float)pf[r >> 3][r & 7];
            if (key < p.T) {
                if (d <= -p.R) lo += pv;
                else if (d >= p.R) up += pv;
                else if (lq < p.T) p.relp[relrow + d + p.R] = (half_t)pv;
            }
        }
        lo += __shfl_xor(lo, 32);
        up += __shfl_xor(up, 32);
        if (hi == 0 && lq < p.T) {
            p.relp[relrow] = (half_t)lo;
            p.relp[relrow + 2 * p.R] = (half_t)up;
        }
    }

    // O^T[d, q] = sum_key V^T[d, key] P^T[key, q]; V^T fragment slot j <-> key 16s + (j&3) + 8*(j>>2) + 4*hi
    __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): this wave's V rows are in LDS (wave-private patch: no barrier)
    f16v oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int d = db * 32 + lq;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h8 vf;
#pragma unroll
            for (int j = 0; j < 8; ++j) vf[j] = sv[(16 * s + (j & 3) + 8 * (j >> 2) + 4 * hi) * VLD + d];
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], s == 0 ? zero16 : oacc[db], 0, 0, 0);
        }
    }
    // Output row (frame lq) of this (pixel, head): column group k = 4 db + gq holds columns 8k + 4 hi .. + 3 in this lane, i.e. a
    // row is split 8 bytes / 8 bytes between lanes l and l + 32.  v_permlane32_swap on the packed words of groups k (vdst) and
    // k + 1 (src) leaves lanes 0-31 with columns 8k .. 8k+7 and lanes 32-63 with 8k+8 .. 8k+15: four 16-byte stores per lane
    // instead of eight 8-byte ones (T21 of the HIP guide; the store tail is issue-bound).
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    u2v pk[8];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            pk[db * 4 + gq] = __builtin_bit_cast(u2v, h4{(half_t)oacc[db][gq * 4 + 0], (half_t)oacc[db][gq * 4 + 1],
                                                        (half_t)oacc[db][gq * 4 + 2], (half_t)oacc[db][gq * 4 + 3]});
    half_t* dst = p.o + (((int64_t)b * p.T + lq) * p.P + pix) * p.ldo + h * 64 + hi * 8;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const unsigned a0 = pk[k][0], a1 = pk[k][1], b0 = pk[k + 1][0], b1 = pk[k + 1][1];
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        if (rvalid) *reinterpret_cast<u4v*>(dst + 8 * k) = u4v{s0[0], s1[0], s0[1], s1[1]};
    }
}

// 32 < T <= 64 frames (no ViewCrafter checkpoint has that many - the UNet's temporal attention would otherwise refuse such a video): the same
// wave-per-(pixel, head) scheme on 2 x 2 score tiles.  Q / K fragment kt / qt: rows 32 kt + lq of the patch; S^T tile (kt, qt) holds keys
// 32 kt + (r&3) + 8 (r>>2) + 4 hi of query 32 qt + lq; the row maximum / sum of a query runs over both key tiles and both lane halves.  The patch is
// 2 x [64][72] halfs per wave (18 KB): one block per CU.  A separate kernel: the T <= 32 one keeps its listing.
template <bool CAUSAL>
__global__ void __launch_bounds__(512) tattn64_d64_kernel(TAttnArgs p) {
    constexpr int VLD = 72;
    constexpr int WAVES = 8;
    __shared__ __attribute__((aligned(16))) half_t sVt[WAVES][2 * 64 * VLD];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    const int64_t pair = (int64_t)blockIdx.x * WAVES + wave;
    if (pair >= p.npairs) return;                                     // wave-uniform; no block-level sync below
    const int h = (int)(pair % p.heads);
    const int64_t pix = (pair / p.heads) % p.P;
    const int b = (int)(pair / (p.heads * p.P));
    const half_t* base = p.qkv + ((int64_t)b * p.T * p.P + pix) * p.ld + h * 64;
    const int64_t fstride = p.P * p.ld;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half_t* sv = sVt[wave];
    half_t* sqk = sv + 64 * VLD;
    h8 qf[2][4], kf[2][4];
    {
        h8 q4[8], k4[8], v4[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (lane >> 3) + 8 * it, ch = lane & 7;
            const half_t* src = base + (int64_t)row * fstride + ch * 8;
            const bool ok = row < p.T;
            q4[it] = ok ? *reinterpret_cast<const h8*>(src) : zero8;
            k4[it] = ok ? *reinterpret_cast<const h8*>(src + p.k_off) : zero8;
            v4[it] = ok ? *reinterpret_cast<const h8*>(src + p.v_off) : zero8;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (lane >> 3) + 8 * it, ch = lane & 7;
            *reinterpret_cast<h8*>(sqk + row * VLD + ch * 8) = q4[it];
            *reinterpret_cast<h8*>(sv + row * VLD + ch * 8) = v4[it];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[t][s] = *reinterpret_cast<const h8*>(sqk + (32 * t + lq) * VLD + s * 16 + hi * 8);
        asm volatile("" ::: "memory");                                // the K rows overwrite the Q rows behind the (in-order) Q fragment reads
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (lane >> 3) + 8 * it, ch = lane & 7;
            *reinterpret_cast<h8*>(sqk + row * VLD + ch * 8) = k4[it];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[t][s] = *reinterpret_cast<const h8*>(sqk + (32 * t + lq) * VLD + s * 16 + hi * 8);
    }
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * 1.4426950408889634f;
    f16v sacc[2][2];                                                  // [key tile][query tile]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            sacc[kt][qt] = zero16;
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][s], qf[qt][s], sacc[kt][qt], 0, 0, 0);
        }
    h8 pf[2][2][2];                                                   // [key tile][query tile][k-step of 16 keys]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = 32 * qt + lq;
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= p.T || (CAUSAL && key > q)) sacc[kt][qt][r] = -1e30f;
                mx = fmaxf(mx, sacc[kt][qt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * c;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][qt][r], c, -mx));
                sacc[kt][qt][r] = e;
                l += e;
            }
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[kt][qt][s][j] = (half_t)(sacc[kt][qt][8 * s + j] * inv);
    }
    // O^T[d, q] = sum_key V^T[d, key] P^T[key, q]: a V^T fragment (d half db, key tile kt, k-step s) serves both query tiles
    f16v oacc[2][2];                                                  // [d half][query tile]
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        oacc[db][0] = zero16;
        oacc[db][1] = zero16;
        const int d = db * 32 + lq;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 vf;
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = sv[(32 * kt + 16 * s + (j & 3) + 8 * (j >> 2) + 4 * hi) * VLD + d];
                oacc[db][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kt][0][s], oacc[db][0], 0, 0, 0);
                oacc[db][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kt][1][s], oacc[db][1], 0, 0, 0);
            }
    }
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        u2v pk[8];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                pk[db * 4 + gq] = __builtin_bit_cast(u2v, h4{(half_t)oacc[db][qt][gq * 4 + 0], (half_t)oacc[db][qt][gq * 4 + 1],
                                                            (half_t)oacc[db][qt][gq * 4 + 2], (half_t)oacc[db][qt][gq * 4 + 3]});
        const int frame = 32 * qt + lq;
        half_t* dst = p.o + (((int64_t)b * p.T + frame) * p.P + pix) * p.ldo + h * 64 + hi * 8;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const unsigned a0 = pk[k][0], a1 = pk[k][1], b0 = pk[k + 1][0], b1 = pk[k + 1][1];
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            if (frame < p.T) *reinterpret_cast<u4v*>(dst + 8 * k) = u4v{s0[0], s1[0], s0[1], s1[1]};
        }
    }
}

// =======================================================================================
// Row softmax in place (fp16 storage, fp32 math), one block per row.
// =======================================================================================
// n need not be a multiple of 8: the columns [n, ceil8(n)) of the last 16-byte chunk take no part in the maximum / sum and are
// written as zeros (the VAE attention at token counts that are not multiples of 8 multiplies this matrix with zero-padded keys).
__global__ void __launch_bounds__(256) softmax_rows_kernel(half_t* x, int n, int64_t ld) {
    __shared__ float red[8];
    half_t* row = x + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    const int nch = (n + 7) >> 3;
    float mx = -1e30f;
    for (int c = tid; c < nch; c += 256) {
        const h8 v = *reinterpret_cast<const h8*>(row + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, c * 8 + e < n ? (float)v[e] : -1e30f);
    }
    mx = vcx_wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = tid; c < nch; c += 256) {
        const h8 v = *reinterpret_cast<const h8*>(row + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += c * 8 + e < n ? __expf((float)v[e] - mx) : 0.f;
    }
    sum = vcx_wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    const float inv = 1.0f / sum;
    for (int c = tid; c < nch; c += 256) {
        h8 v = *reinterpret_cast<const h8*>(row + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c * 8 + e < n ? (half_t)(__expf((float)v[e] - mx) * inv) : (half_t)0.f;
        *reinterpret_cast<h8*>(row + c * 8) = v;
    }
}

}  // namespace

extern "C" int vcx_attn_flash_d64_f16(const void* q, const void* k, const void* vt, void* o, int n_groups, int heads,
                                      int nq, int nk, int kv_rows, int kv_div, int64_t ldq, int64_t ldk, int64_t ldvt,
                                      int64_t ldo, float scale, int flags, void* stream) {
    VCX_REQUIRE(q && k && vt && o, "vcx_attn_flash_d64_f16: null pointer");
    VCX_REQUIRE(n_groups > 0 && heads > 0 && nq > 0 && nk > 0 && kv_div > 0, "vcx_attn_flash_d64_f16: empty problem");
    VCX_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && kv_rows % 8 == 0 && kv_rows >= nk,
                "vcx_attn_flash_d64_f16: strides must be multiples of 8 and kv_rows >= nk (ldq=%lld ldk=%lld ldvt=%lld kv_rows=%d nk=%d)",
                (long long)ldq, (long long)ldk, (long long)ldvt, kv_rows, nk);
    VCX_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0 && ((uintptr_t)o & 7) == 0,
                "vcx_attn_flash_d64_f16: pointers must be 16-byte aligned");
    VCX_REQUIRE((int64_t)n_groups * heads * ((nq + 127) / 128) < (1ll << 30), "vcx_attn_flash_d64_f16: too many workgroups");
    FlashArgs a;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.vt = (const half_t*)vt; a.o = (half_t*)o;
    a.heads = heads; a.nq = nq; a.nk = nk; a.kv_rows = kv_rows; a.kv_div = kv_div;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.accumulate = flags;
    a.k2 = nullptr; a.vt2 = nullptr; a.nk2 = 0; a.kv_rows2 = 0; a.kv_div2 = 1; a.ldk2 = 0; a.ldvt2 = 0;
    const bool pre = flags & VCX_ATTN_LOG2_LOGITS;
    hipStream_t s = (hipStream_t)stream;
    const double nprob = (double)n_groups * heads;
    VcxProfScope prof(VCX_FAM_FLASH, s, 4.0 * nprob * nq * (double)nk * 64, 2.0 * nprob * 64 * (2.0 * nq + 2.0 * nk));
    VCX_REQUIRE(((int64_t)(nk - 1) * ldk + 64) * 2 < 0xFFFF0000ll && (63ll * ldvt + nk + 8) * 2 < 0xFFFF0000ll,
                "vcx_attn_flash_d64_f16: K / V^T extents per (group, head) must stay below 4 GiB");
    // Long key sequences with base-2 logits and whole 64-key tiles: the software-pipelined kernel (attention_v2.hip: MFMA and
    // softmax overlapped inside one wave per SIMD).  One block per CU and a prologue that is not hidden behind another block:
    // it pays from ~2000 keys up (same-box A/Bs: round 3, profiles/r03_flash_v2.md: +4.5 % at 9216 keys, level at 2304 / 1152, -18 % at 576;
    // round 6 with the K rows permuted - no half swaps in the softmax stream - profiles/r06a_flash_variants_ab.txt, r06g_flash_sumv_ab.txt:
    // +5 ... +7 % at 2304 keys on two boxes, level ... +5 % at 1152, -15 % at 576; knob FLASH_IMPL: 1 = never, 2 = whenever the shapes allow)
    const int impl = vcx_tune(VCX_TUNE_FLASH_IMPL);
    if (impl != 1 && pre && !(flags & VCX_ATTN_ACCUMULATE) && nk % 64 == 0 && (impl == 2 || nk >= 2048)) {
        Flash2Args f;
        f.q = (const half_t*)q; f.k = (const half_t*)k; f.vt = (const half_t*)vt; f.o = (half_t*)o;
        f.heads = heads; f.nq = nq; f.nk = nk; f.kv_rows = kv_rows; f.kv_div = kv_div;
        f.ldq = ldq; f.ldk = ldk; f.ldvt = ldvt; f.ldo = ldo;
        f.nqb = (nq + 255) / 256;
        f.nprob = n_groups * heads;
        return vcx_flash2_launch(f, s);
    }
    // two 32-row query blocks per wave (256 rows per block) unless the row count would waste > 20 % of such blocks
    const int force_qb = vcx_tune(VCX_TUNE_FLASH_QB);
    const int blocks2 = (nq + 255) / 256;
    bool qb2 = (double)nq / (blocks2 * 256.0) >= 0.8;
    if (force_qb == 1) qb2 = false;
    if (force_qb == 2) qb2 = true;
    a.nprob = n_groups * heads;
    const int prob_pad = (a.nprob + 7) / 8 * 8;
    if (qb2) {
        a.nqb = blocks2;
        dim3 grid(blocks2 * prob_pad);
        if (pre) hipLaunchKernelGGL((flash_d64_kernel<2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((flash_d64_kernel<2, false>), grid, dim3(256), 0, s, a);
    } else {
        a.nqb = (nq + 127) / 128;
        dim3 grid(a.nqb * prob_pad);
        if (pre) hipLaunchKernelGGL((flash_d64_kernel<1, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((flash_d64_kernel<1, false>), grid, dim3(256), 0, s, a);
    }
    return vcx_check_launch("vcx_attn_flash_d64_f16");
}

extern "C" int vcx_attn_flash_dual_d64_f16(const void* q, const void* k1, const void* vt1, const void* k2, const void* vt2, void* o,
                                           int n_groups, int heads, int nq, int nk1, int kv_rows1, int kv_div1, int64_t ldk1,
                                           int64_t ldvt1, int nk2, int kv_rows2, int kv_div2, int64_t ldk2, int64_t ldvt2,
                                           int64_t ldq, int64_t ldo, float scale, int flags, void* stream) {
    VCX_REQUIRE(q && k1 && vt1 && k2 && vt2 && o, "vcx_attn_flash_dual_d64_f16: null pointer");
    VCX_REQUIRE(n_groups > 0 && heads > 0 && nq > 0 && nk1 > 0 && nk2 > 0 && kv_div1 > 0 && kv_div2 > 0,
                "vcx_attn_flash_dual_d64_f16: empty problem");
    VCX_REQUIRE(ldq % 8 == 0 && ldo % 4 == 0 && ldk1 % 8 == 0 && ldvt1 % 8 == 0 && ldk2 % 8 == 0 && ldvt2 % 8 == 0 &&
                kv_rows1 % 8 == 0 && kv_rows1 >= nk1 && kv_rows2 % 8 == 0 && kv_rows2 >= nk2,
                "vcx_attn_flash_dual_d64_f16: strides must be multiples of 8 and kv_rows >= nk");
    VCX_REQUIRE((((uintptr_t)q | (uintptr_t)k1 | (uintptr_t)vt1 | (uintptr_t)k2 | (uintptr_t)vt2) & 15) == 0 && ((uintptr_t)o & 7) == 0,
                "vcx_attn_flash_dual_d64_f16: pointers must be 16-byte aligned");
    VCX_REQUIRE(!(flags & VCX_ATTN_ACCUMULATE), "vcx_attn_flash_dual_d64_f16: VCX_ATTN_ACCUMULATE is not supported here");
    VCX_REQUIRE((int64_t)n_groups * heads * ((nq + 127) / 128) < (1ll << 30), "vcx_attn_flash_dual_d64_f16: too many workgroups");
    VCX_REQUIRE(((int64_t)(nk1 - 1) * ldk1 + 64) * 2 < 0xFFFF0000ll && (63ll * ldvt1 + nk1 + 8) * 2 < 0xFFFF0000ll &&
                ((int64_t)(nk2 - 1) * ldk2 + 64) * 2 < 0xFFFF0000ll && (63ll * ldvt2 + nk2 + 8) * 2 < 0xFFFF0000ll,
                "vcx_attn_flash_dual_d64_f16: K / V^T extents per (group, head) must stay below 4 GiB");
    FlashArgs a;
    a.q = (const half_t*)q; a.o = (half_t*)o;
    a.k = (const half_t*)k1; a.vt = (const half_t*)vt1; a.nk = nk1; a.kv_rows = kv_rows1; a.kv_div = kv_div1; a.ldk = ldk1; a.ldvt = ldvt1;
    a.k2 = (const half_t*)k2; a.vt2 = (const half_t*)vt2; a.nk2 = nk2; a.kv_rows2 = kv_rows2; a.kv_div2 = kv_div2; a.ldk2 = ldk2; a.ldvt2 = ldvt2;
    a.heads = heads; a.nq = nq; a.ldq = ldq; a.ldo = ldo;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.accumulate = flags;
    a.nprob = n_groups * heads;
    hipStream_t s = (hipStream_t)stream;
    const double nprob = (double)n_groups * heads;
    VcxProfScope prof(VCX_FAM_FLASH, s, 4.0 * nprob * nq * (double)(nk1 + nk2) * 64, 2.0 * nprob * 64 * (2.0 * nq + 2.0 * (nk1 + nk2)));
    // Both key sets shared by the same frame groups and small enough for LDS: the resident-K/V kernel (no per-tile pipeline)
    const bool resident_on = vcx_tune(VCX_TUNE_XATTN_RESIDENT) != 0;
    if (resident_on && kv_div1 == kv_div2 && n_groups % kv_div1 == 0 && nk1 <= 128 && nk2 <= 256) {
        XAttnArgs x;
        x.q = (const half_t*)q; x.o = (half_t*)o;
        x.k1 = (const half_t*)k1; x.vt1 = (const half_t*)vt1; x.k2 = (const half_t*)k2; x.vt2 = (const half_t*)vt2;
        x.heads = heads; x.nk1 = nk1; x.nk2 = nk2; x.kv_rows1 = kv_rows1; x.kv_rows2 = kv_rows2;
        x.ldq = ldq; x.ldo = ldo; x.ldk1 = ldk1; x.ldvt1 = ldvt1; x.ldk2 = ldk2; x.ldvt2 = ldvt2;
        x.scale_log2 = (flags & VCX_ATTN_LOG2_LOGITS) ? 1.0f : scale * 1.4426950408889634f;
        x.nunits = (n_groups / kv_div1) * heads;
        x.rows_per_unit = (int64_t)kv_div1 * nq;
        static const int cus = []() {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                return prop.multiProcessorCount;
            return 256;
        }();
        int split = cus / x.nunits;
        if (split < 1) split = 1;
        const int64_t iters = (x.rows_per_unit + 511) / 512;
        if (split > iters) split = (int)iters;
        x.rows_per_block = ((iters + split - 1) / split) * 512;
        x.split = (int)((x.rows_per_unit + x.rows_per_block - 1) / x.rows_per_block);
        constexpr int XSMEM = 6 * 64 * 64 * 2 * 2;
        // second form (half tiles, prefetched fragments and query rows, first result parked in the other 64 KB of LDS): needs 32-bit
        // byte offsets inside a unit's Q / O rows and 16-byte output stores; knob XATTN_RESIDENT = 2 keeps the first form (A/B runs)
        const int64_t q_ext = ((x.rows_per_unit - 1) * ldq + 64) * 2, o_ext = ((x.rows_per_unit - 1) * ldo + 64) * 2;
        if (vcx_tune(VCX_TUNE_XATTN_RESIDENT) != 2 && q_ext < 0xFFFF0000ll && o_ext < 0xFFFF0000ll && ldo % 8 == 0 && ((uintptr_t)o & 15) == 0) {
            constexpr int XSMEM2 = XSMEM + 8 * 8192;
            static VcxLdsAttr lds2;
            if (!lds2.ensure(reinterpret_cast<const void*>(xattn_resident2_d64_kernel), XSMEM2, "vcx_attn_flash_dual_d64_f16(resident2)")) return VCX_ELAUNCH;
            hipLaunchKernelGGL(xattn_resident2_d64_kernel, dim3(x.nunits * x.split), dim3(512), XSMEM2, s, x, (unsigned)q_ext, (unsigned)o_ext);
            return vcx_check_launch("vcx_attn_flash_dual_d64_f16(resident2)");
        }
        static VcxLdsAttr lds;
        if (!lds.ensure(reinterpret_cast<const void*>(xattn_resident_d64_kernel), XSMEM, "vcx_attn_flash_dual_d64_f16")) return VCX_ELAUNCH;
        hipLaunchKernelGGL(xattn_resident_d64_kernel, dim3(x.nunits * x.split), dim3(512), XSMEM, s, x);
        return vcx_check_launch("vcx_attn_flash_dual_d64_f16(resident)");
    }
    // two query blocks per wave (half the LDS fragment traffic per MFMA) unless that wastes > 20 % of the 256-row blocks; with
    // two blocks the kept first result is packed fp16 and the running-max-in-C variant is not used (register budget)
    const int force_qb = vcx_tune(VCX_TUNE_FLASH_QB);
    const int blocks2 = (nq + 255) / 256;
    bool qb2 = (double)nq / (blocks2 * 256.0) >= 0.8;
    if (force_qb == 1) qb2 = false;
    if (force_qb == 2) qb2 = true;
    const bool pre = flags & VCX_ATTN_LOG2_LOGITS;
    a.nqb = qb2 ? blocks2 : (nq + 127) / 128;
    dim3 grid(a.nqb * ((a.nprob + 7) / 8 * 8));
    if (qb2) {
        if (pre) a.scale_log2 = 1.0f;       // the plain kernel multiplies the scores by scale_log2: base-2 logits need 1
        hipLaunchKernelGGL((flash_d64_kernel<2, false, true>), grid, dim3(256), 0, s, a);
    } else if (pre) {
        hipLaunchKernelGGL((flash_d64_kernel<1, true, true>), grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((flash_d64_kernel<1, false, true>), grid, dim3(256), 0, s, a);
    }
    return vcx_check_launch("vcx_attn_flash_dual_d64_f16");
}

extern "C" int vcx_attn_flash_d512_f16(const void* q, const void* k, const void* vt, void* o, int n_groups, int nq, int nk, int kv_rows, int64_t ldq,
                                       int64_t ldk, int64_t ldvt, int64_t ldo, float scale, void* stream) {
    VCX_REQUIRE(q && k && vt && o, "vcx_attn_flash_d512_f16: null pointer");
    VCX_REQUIRE(n_groups > 0 && nq > 0 && nk > 0, "vcx_attn_flash_d512_f16: empty problem");
    VCX_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && kv_rows % 8 == 0 && kv_rows >= nk,
                "vcx_attn_flash_d512_f16: strides must be multiples of 8 and kv_rows >= nk (ldq=%lld ldk=%lld ldvt=%lld kv_rows=%d nk=%d)",
                (long long)ldq, (long long)ldk, (long long)ldvt, kv_rows, nk);
    VCX_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0 && ((uintptr_t)o & 7) == 0, "vcx_attn_flash_d512_f16: pointers must be 16-byte aligned");
    VCX_REQUIRE(((int64_t)(nk - 1) * ldk + 512) * 2 < 0xFFFF0000ll && (511ll * ldvt + nk + 8) * 2 < 0xFFFF0000ll,
                "vcx_attn_flash_d512_f16: K / V^T extents per group must stay below 4 GiB");
    Flash512Args a;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.vt = (const half_t*)vt; a.o = (half_t*)o;
    a.nq = nq; a.nk = nk; a.kv_rows = kv_rows; a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.nqb = (nq + 127) / 128;
    a.nprob = n_groups;
    VCX_REQUIRE((int64_t)a.nqb * ((n_groups + 7) / 8 * 8) < (1ll << 30), "vcx_attn_flash_d512_f16: too many workgroups");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_FLASH, s, 4.0 * n_groups * (double)nq * nk * 512, 2.0 * n_groups * 512 * (2.0 * nq + 2.0 * nk));
    constexpr int FSMEM = 2 * 2 * 32 * 512 * 2;
    static VcxLdsAttr lds;
    if (!lds.ensure(reinterpret_cast<const void*>(flash_d512_kernel), FSMEM, "vcx_attn_flash_d512_f16")) return VCX_ELAUNCH;
    hipLaunchKernelGGL(flash_d512_kernel, dim3((unsigned)(a.nqb * ((n_groups + 7) / 8 * 8))), dim3(512), FSMEM, s, a);
    return vcx_check_launch("vcx_attn_flash_d512_f16");
}

extern "C" int vcx_attn_temporal_d64_masked_f16(const void* qkv, void* o, int B, int T, int64_t P, int heads, int64_t ld,
                                                int k_off, int v_off, int64_t ldo, float scale, int flags, void* stream);
extern "C" int vcx_attn_temporal_d64_f16(const void* qkv, void* o, int B, int T, int64_t P, int heads, int64_t ld,
                                         int k_off, int v_off, int64_t ldo, float scale, void* stream) {
    return vcx_attn_temporal_d64_masked_f16(qkv, o, B, T, P, heads, ld, k_off, v_off, ldo, scale, 0, stream);
}

static int tattn_launch(const void* qkv, void* o, const void* relg, void* relp, int R, int B, int T, int64_t P, int heads, int64_t ld,
                        int k_off, int v_off, int64_t ldo, float scale, int flags, void* stream);

extern "C" int vcx_attn_temporal_d64_masked_f16(const void* qkv, void* o, int B, int T, int64_t P, int heads, int64_t ld,
                                                int k_off, int v_off, int64_t ldo, float scale, int flags, void* stream) {
    return tattn_launch(qkv, o, nullptr, nullptr, 0, B, T, P, heads, ld, k_off, v_off, ldo, scale, flags, stream);
}

extern "C" int vcx_attn_temporal_d64_rel_f16(const void* qkv, void* o, const void* relg, void* relp, int R, int B, int T, int64_t P, int heads,
                                             int64_t ld, int k_off, int v_off, int64_t ldo, float scale, int flags, void* stream) {
    VCX_REQUIRE(relg && relp && R >= 1 && 2 * R + 1 <= 64 && T <= 32 && (((uintptr_t)relg | (uintptr_t)relp) & 1) == 0,
                "vcx_attn_temporal_d64_rel_f16: need relg / relp, 1 <= R <= 31 (2R + 1 distances in 64 slots), T <= 32 (R=%d T=%d)", R, T);
    return tattn_launch(qkv, o, relg, relp, R, B, T, P, heads, ld, k_off, v_off, ldo, scale, flags, stream);
}

static int tattn_launch(const void* qkv, void* o, const void* relg, void* relp, int R, int B, int T, int64_t P, int heads, int64_t ld,
                        int k_off, int v_off, int64_t ldo, float scale, int flags, void* stream) {
    VCX_REQUIRE(qkv && o, "vcx_attn_temporal_d64_f16: null pointer");
    VCX_REQUIRE((flags & ~VCX_ATTN_CAUSAL) == 0, "vcx_attn_temporal_d64_masked_f16: unknown flags 0x%x (VCX_ATTN_CAUSAL is the only one)", flags);
    VCX_REQUIRE(B > 0 && T > 0 && T <= 64 && P > 0 && heads > 0, "vcx_attn_temporal_d64_f16: need 0 < T <= 64 (T=%d)", T);
    VCX_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0,
                "vcx_attn_temporal_d64_f16: strides/offsets must be multiples of 8");
    VCX_REQUIRE((((uintptr_t)qkv | (uintptr_t)o) & 15) == 0, "vcx_attn_temporal_d64_f16: pointers must be 16-byte aligned");
    TAttnArgs a;
    a.qkv = (const half_t*)qkv; a.o = (half_t*)o;
    a.B = B; a.T = T; a.heads = heads; a.P = P; a.ld = ld; a.ldo = ldo;
    a.k_off = k_off; a.v_off = v_off; a.scale = scale;
    a.npairs = (int64_t)B * P * heads;
    a.relg = (const half_t*)relg; a.relp = (half_t*)relp; a.R = R;
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_TATTN, s, 4.0 * a.npairs * (double)T * T * 64, 2.0 * a.npairs * T * 64 * 4.0);
    const int64_t nblk = (a.npairs + 7) / 8;
    VCX_REQUIRE(nblk < (1ll << 31), "vcx_attn_temporal_d64_f16: grid too large");
    if (T > 32) {         // 2 x 2 score tiles, 18 KB of LDS per wave
        if (flags & VCX_ATTN_CAUSAL) hipLaunchKernelGGL(tattn64_d64_kernel<true>, dim3((unsigned)nblk), dim3(512), 0, s, a);
        else hipLaunchKernelGGL(tattn64_d64_kernel<false>, dim3((unsigned)nblk), dim3(512), 0, s, a);
    } else if (relg) {
        if (flags & VCX_ATTN_CAUSAL) hipLaunchKernelGGL((tattn_d64_kernel<true, true>), dim3((unsigned)nblk), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((tattn_d64_kernel<false, true>), dim3((unsigned)nblk), dim3(512), 0, s, a);
    } else if (flags & VCX_ATTN_CAUSAL) hipLaunchKernelGGL(tattn_d64_kernel<true>, dim3((unsigned)nblk), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(tattn_d64_kernel<false>, dim3((unsigned)nblk), dim3(512), 0, s, a);
    return vcx_check_launch("vcx_attn_temporal_d64_f16");
}

extern "C" int vcx_softmax_rows_f16(void* x, int64_t rows, int n, int64_t ld, void* stream) {
    VCX_REQUIRE(x && rows > 0 && n > 0, "vcx_softmax_rows_f16: empty problem");
    VCX_REQUIRE(ld % 8 == 0 && ld >= (n + 7) / 8 * 8 && ((uintptr_t)x & 15) == 0,
                "vcx_softmax_rows_f16: ld must be a multiple of 8 that covers n rounded up to 8 (n=%d ld=%lld), x 16-byte aligned", n, (long long)ld);
    VCX_REQUIRE(rows < (1ll << 31), "vcx_softmax_rows_f16: too many rows");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 4.0 * rows * (double)n);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, (half_t*)x, n, ld);
    return vcx_check_launch("vcx_softmax_rows_f16");
}
