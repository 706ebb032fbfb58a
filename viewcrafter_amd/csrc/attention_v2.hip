// Flash attention, d = 64, software-pipelined inside ONE wave per SIMD (gfx950).
//
// Why a second kernel.  flash_d64_kernel (attention.hip) is phased: per 64-key tile a wave issues 16 score MFMAs, then ~230
// VALU instructions of softmax, then 16 PV MFMAs - on this part the VALU stream of a wave does not run under the MFMAs of the
// OTHER wave of the SIMD (profiles/r01_gemm_experiments.md, tools/ubench.hip), so matrix-pipe time and softmax time add
// (~2500 cycles per tile for 1024 cycles of MFMA).  What does overlap is a wave's own VALU stream with its own MFMAs: up to ~5
// single-issue instructions fit into the 32-cycle shadow of a v_mfma_f32_32x32x16 (MI355X_MICROARCH.md).  This kernel gives
// every MFMA such a shadow: while the score MFMAs of tile j+1 run, the wave exponentiates tile j; while the PV MFMAs of tile j
// run, it finishes that and takes the row maxima of tile j+1.  That needs both score tiles live (2 x 64 registers) next to O
// (64), Q (32) and the running-max C tuples (32): more than 256 registers, i.e. one wave per SIMD and the 512-entry file.
//
// Register files by construction, not by the allocator's choice: with a 512-register budget hipcc selects the AGPR form for
// every MFMA builtin and then shuttles the score tiles through v_accvgpr_read (64 extra VALU per tile; the 1-wave-per-SIMD
// variants of rounds 1-2 died of exactly that), and an "a"-constrained asm operand that is loop-carried gets its live range
// split through VGPRs (64 copies in, 64 out, per tile).  So the accumulator half of the file is owned by name: O^T lives in
// a[0:63] (query block b, d half db: a[32b + 16db ...]), the Q fragments in a[64:95] (block b, k-step s: a[64 + 16b + 4s ...]);
// every statement that writes them lists them as clobbers (which also makes the kernel descriptor allocate them), nothing
// else may touch AGPRs - the build audit (tools/isa_scan.py flash2) requires that no compiler-generated v_accvgpr_* and no
// scratch exist in this kernel.  Score tiles, K / V^T / P fragments are ordinary compiler-allocated VGPR values ("v" operands).
// What the compiler cannot see inside an asm statement, and how it is covered:
//   * VALU-written VGPR -> MFMA source (P from v_cvt_pk / v_permlane, the C tuple after a rescale): the string starts with s_nop 1;
//   * MFMA result -> VALU read: the first reader of a score tile is >= 100 instructions behind its last MFMA by schedule
//     (row maxima start ten MFMAs later); O is only read behind an explicit s_nop 15 x 2 (rescale branch, epilogue);
//   * accumulate chains use the same tuple as C and D (no wait states needed);
//   * v_cvt_pk -> v_permlane32_swap: >= 2 instructions apart by the order of the softmax stream.
//
// Pipeline of one block (4 waves x 64 query rows, 64-key tiles, K two tiles and V^T one tile ahead by LDS-DMA into two
// compile-time addressed slots each - distinct __shared__ objects, so the compiler's LDS-DMA alias tracking waits for a DMA
// only in front of reads of the SAME slot; with a run-time slot index it drains the prefetch at every tile):
//   iteration j:  gaps  0-15  S(j+1) = K(j+1) Q^T      | exp2 / row sums / fp16 packing / half swaps of tile j
//                 mid         vmcnt(0), s_barrier; first V^T(j) fragments, DMA K(j+3), a burst of the softmax stream
//                 gaps 16-31  O += V^T(j) P(j)         | rest of tile j, DMA V^T(j+1), row maxima of tile j+1, max decision
// (one barrier per tile: V^T(j), issued at mid(j-1), is complete for every wave behind mid(j) - not earlier - and K(j+2) too)
// The loop is unrolled by two so that slot addresses and the two score-tile register sets are compile-time constants.
//
// Measured (MI355X, 50 frames x 5 heads x 9216^2, profiles/r03_flash_v2.md): 977 TF/s against 935 for the phased kernel on the same
// box; timing-only ablations of the key loop (tools/flash_ablate.py) put the bare MFMA stream of a tile at 0.72 us = 1.49 PFLOP/s
// - the matrix pipe issues back to back and the board's power cap sets the clock (~1.4 GHz) - and what is NOT hidden under it at
// DMA issue 0.10 us, LDS fragment reads 0.07, softmax stream 0.2, barrier 0.03 per tile.  Row sums as 8 extra MFMAs per tile
// (SUMV = 0: 64 fewer VALU instructions) lose 4 % to the plain adds: the pipe, not instruction issue, is the scarce resource.
#include "flash2.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32;
typedef u32 u4v __attribute__((ext_vector_type(4)));
[[maybe_unused]] constexpr float F2_DEFER = 8.0f;
#define F2_VAR_DEFAULT 1                  // stream variant (Stream<SUMV, VAR>): 1 no half swaps (permuted K rows) | 2 paired row sums
#define F2_SUMV_DEFAULT 1                 // row sums: 0 = on the matrix pipe, 1 = v_add_f32 in the softmax stream          // log2 units, as FLASH_DEFER in attention.hip
#define FI __device__ __forceinline__

FI int tile_off2(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <int I, int N, class F>
FI void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// ---- the asm-owned accumulator file
#define CLOB_O "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",  \
               "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37",    \
               "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",    \
               "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define CLOB_L "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112",   \
               "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128",  \
               "a129", "a130", "a131"
#define CLOB_Q "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81",   \
               "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"

// score MFMAs: D(v) = K(v) x Q(a[...]) + C.  I = 4 b + s selects the Q fragment.
template <int I>
FI void mfma_qk_zero(f16v& d, const h8& kf) {
#define VCX_QK0(n, lo, hi) if constexpr (I == n) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[" #lo ":" #hi "], 0" : "=&v"(d) : "v"(kf))
    VCX_QK0(0, 64, 67); VCX_QK0(1, 68, 71); VCX_QK0(2, 72, 75); VCX_QK0(3, 76, 79);
    VCX_QK0(4, 80, 83); VCX_QK0(5, 84, 87); VCX_QK0(6, 88, 91); VCX_QK0(7, 92, 95);
#undef VCX_QK0
}
template <int I>
FI void mfma_qk_first(f16v& d, const h8& kf, const f16v& c) {       // C = -running max in all slots
#define VCX_QK1(n, lo, hi) if constexpr (I == n) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[" #lo ":" #hi "], %2" : "=&v"(d) : "v"(kf), "v"(c))
    VCX_QK1(0, 64, 67); VCX_QK1(1, 68, 71); VCX_QK1(2, 72, 75); VCX_QK1(3, 76, 79);
    VCX_QK1(4, 80, 83); VCX_QK1(5, 84, 87); VCX_QK1(6, 88, 91); VCX_QK1(7, 92, 95);
#undef VCX_QK1
}
template <int I>
FI void mfma_qk_acc(f16v& d, const h8& kf) {
#define VCX_QK2(n, lo, hi) if constexpr (I == n) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[" #lo ":" #hi "], %0" : "+v"(d) : "v"(kf))
    VCX_QK2(0, 64, 67); VCX_QK2(1, 68, 71); VCX_QK2(2, 72, 75); VCX_QK2(3, 76, 79);
    VCX_QK2(4, 80, 83); VCX_QK2(5, 84, 87); VCX_QK2(6, 88, 91); VCX_QK2(7, 92, 95);
#undef VCX_QK2
}
// PV MFMAs: O^T(a) += V^T(v) x P(v).  I = 2 b + db selects the accumulator.  No clobber list and no s_nop on the hot statements:
// hipcc pads an s_nop in front of an asm statement that clobbers registers, and P is written >= 2 instructions ahead by
// schedule (tools/isa_audit.py check 7); the statements that initialise / rescale / read the accumulators carry the clobbers.
template <int I>
FI void mfma_pv(const h8& vf, const u4v& pf) {
#define VCX_PV(n, lo, hi) if constexpr (I == n) asm volatile("v_mfma_f32_32x32x16_f16 a[" #lo ":" #hi "], %0, %1, a[" #lo ":" #hi "]" : : "v"(vf), "v"(pf))
    VCX_PV(0, 0, 15); VCX_PV(1, 16, 31); VCX_PV(2, 32, 47); VCX_PV(3, 48, 63);
#undef VCX_PV
}
// Row sums on the matrix pipe: L_b(a) += ONES(a) x P(v) - every row of the 32 x 32 result is sum_k P[k][q], over all 16 keys of
// the k-step (both lane halves).  8 MFMAs per tile instead of 64 v_add_f32: the key loop is bound by instruction issue (one wave
// per SIMD issues ~1 instruction per 5 cycles), not by the pipe, and the sum is taken over the fp16 P that the PV product uses.
template <int B>
FI void mfma_rowsum(const u4v& pf) {
    if constexpr (B == 0) asm volatile("v_mfma_f32_32x32x16_f16 a[96:111], a[128:131], %0, a[96:111]" : : "v"(pf));
    else asm volatile("v_mfma_f32_32x32x16_f16 a[112:127], a[128:131], %0, a[112:127]" : : "v"(pf));
}
FI void acc_zero_o(const u32& ones) {       // ones = 0x3c003c00: two fp16 1.0
#define Z4(a, b, c, d) "v_accvgpr_write_b32 a" #a ", 0\n\tv_accvgpr_write_b32 a" #b ", 0\n\tv_accvgpr_write_b32 a" #c ", 0\n\tv_accvgpr_write_b32 a" #d ", 0\n\t"
    asm volatile(Z4(0, 1, 2, 3) Z4(4, 5, 6, 7) Z4(8, 9, 10, 11) Z4(12, 13, 14, 15) Z4(16, 17, 18, 19) Z4(20, 21, 22, 23) Z4(24, 25, 26, 27)
                 Z4(28, 29, 30, 31) Z4(32, 33, 34, 35) Z4(36, 37, 38, 39) Z4(40, 41, 42, 43) Z4(44, 45, 46, 47) Z4(48, 49, 50, 51)
                 Z4(52, 53, 54, 55) Z4(56, 57, 58, 59) Z4(60, 61, 62, 63) "s_nop 1" : : : CLOB_O);
    asm volatile(Z4(96, 97, 98, 99) Z4(100, 101, 102, 103) Z4(104, 105, 106, 107) Z4(108, 109, 110, 111) Z4(112, 113, 114, 115) Z4(116, 117, 118, 119)
                 Z4(120, 121, 122, 123) Z4(124, 125, 126, 127)
                 "v_accvgpr_write_b32 a128, %0\n\tv_accvgpr_write_b32 a129, %0\n\tv_accvgpr_write_b32 a130, %0\n\tv_accvgpr_write_b32 a131, %0\n\ts_nop 1"
                 : : "v"(ones) : CLOB_L);
#undef Z4
}
// Q fragment I = 4 b + s (four dwords) into a[64 + 4 I ...]
template <int I>
FI void acc_load_q(const u4v& w) {
#define VCX_LQ(n, r0, r1, r2, r3) if constexpr (I == n) asm volatile("v_accvgpr_write_b32 a" #r0 ", %0\n\tv_accvgpr_write_b32 a" #r1 ", %1\n\tv_accvgpr_write_b32 a" #r2 ", %2\n\tv_accvgpr_write_b32 a" #r3 ", %3\n\ts_nop 1" : : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : CLOB_Q)
    VCX_LQ(0, 64, 65, 66, 67); VCX_LQ(1, 68, 69, 70, 71); VCX_LQ(2, 72, 73, 74, 75); VCX_LQ(3, 76, 77, 78, 79);
    VCX_LQ(4, 80, 81, 82, 83); VCX_LQ(5, 84, 85, 86, 87); VCX_LQ(6, 88, 89, 90, 91); VCX_LQ(7, 92, 93, 94, 95);
#undef VCX_LQ
}
// O^T of query block B (a[32 B .. 32 B + 31]) and its row sums L_B (a[96 + 16 B ...]) times alpha (per lane), in place; the
// MFMAs in flight complete first
#define S1(n) "v_accvgpr_read_b32 %0, a" #n "\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a" #n ", %0\n\t"
template <int B>
FI void acc_scale_o(const float& alpha) {
    float t;
    if constexpr (B == 0)
        asm volatile("s_nop 15\n\ts_nop 15\n\t" S1(0) S1(1) S1(2) S1(3) S1(4) S1(5) S1(6) S1(7) S1(8) S1(9) S1(10) S1(11) S1(12) S1(13) S1(14) S1(15)
                     S1(16) S1(17) S1(18) S1(19) S1(20) S1(21) S1(22) S1(23) S1(24) S1(25) S1(26) S1(27) S1(28) S1(29) S1(30) S1(31)
                     S1(96) S1(97) S1(98) S1(99) S1(100) S1(101) S1(102) S1(103) S1(104) S1(105) S1(106) S1(107) S1(108) S1(109) S1(110) S1(111) "s_nop 1"
                     : "=&v"(t) : "v"(alpha) : CLOB_O, CLOB_L);
    else
        asm volatile("s_nop 15\n\ts_nop 15\n\t" S1(32) S1(33) S1(34) S1(35) S1(36) S1(37) S1(38) S1(39) S1(40) S1(41) S1(42) S1(43) S1(44) S1(45) S1(46) S1(47)
                     S1(48) S1(49) S1(50) S1(51) S1(52) S1(53) S1(54) S1(55) S1(56) S1(57) S1(58) S1(59) S1(60) S1(61) S1(62) S1(63)
                     S1(112) S1(113) S1(114) S1(115) S1(116) S1(117) S1(118) S1(119) S1(120) S1(121) S1(122) S1(123) S1(124) S1(125) S1(126) S1(127) "s_nop 1"
                     : "=&v"(t) : "v"(alpha) : CLOB_O, CLOB_L);
}
#undef S1
// accumulator I = 2 b + db -> 16 VGPR floats (epilogue)
template <int I>
FI void acc_read_o(f16v& o) {
#define R4(k, a, b, c, d) "v_accvgpr_read_b32 %" #k ", a" #a "\n\t" "v_accvgpr_read_b32 %" #b ", a" #c "\n\t"
#define VCX_RD(n, r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15)                                                               \
    if constexpr (I == n)                                                                                                                             \
        asm volatile("s_nop 15\n\ts_nop 15\n\t"                                                                                                       \
                     "v_accvgpr_read_b32 %0, a" #r0 "\n\tv_accvgpr_read_b32 %1, a" #r1 "\n\tv_accvgpr_read_b32 %2, a" #r2 "\n\tv_accvgpr_read_b32 %3, a" #r3 "\n\t"    \
                     "v_accvgpr_read_b32 %4, a" #r4 "\n\tv_accvgpr_read_b32 %5, a" #r5 "\n\tv_accvgpr_read_b32 %6, a" #r6 "\n\tv_accvgpr_read_b32 %7, a" #r7 "\n\t"    \
                     "v_accvgpr_read_b32 %8, a" #r8 "\n\tv_accvgpr_read_b32 %9, a" #r9 "\n\tv_accvgpr_read_b32 %10, a" #r10 "\n\tv_accvgpr_read_b32 %11, a" #r11 "\n\t" \
                     "v_accvgpr_read_b32 %12, a" #r12 "\n\tv_accvgpr_read_b32 %13, a" #r13 "\n\tv_accvgpr_read_b32 %14, a" #r14 "\n\tv_accvgpr_read_b32 %15, a" #r15 "\n\ts_nop 1" \
                     : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]), "=v"(o[8]), "=v"(o[9]),     \
                       "=v"(o[10]), "=v"(o[11]), "=v"(o[12]), "=v"(o[13]), "=v"(o[14]), "=v"(o[15]))
    VCX_RD(0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    VCX_RD(1, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
    VCX_RD(2, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47);
    VCX_RD(3, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63);
#undef VCX_RD
#undef R4
}

template <int B>
FI float acc_read_l() {     // every element of L_B is the row sum of this lane's query
    float l;
    if constexpr (B == 0) asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a96\n\ts_nop 1" : "=v"(l));
    else asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a112\n\ts_nop 1" : "=v"(l));
    return l;
}

// ---- softmax VALU as asm statements: `asm volatile` keeps them exactly where the schedule below puts them (pure HIP arithmetic
// is placed by instruction selection next to its consumer, whatever sched_barrier says) and exactly these opcodes (no
// v_pk_add_f32 from the SLP vectoriser, no canonicalising v_max in front of fmaxf: both measured anti-levers beside MFMAs).
// (macros, not functions: an element of a register tuple cannot bind to a reference.)  Two rules of the stream below:
//   * consecutive statements alternate between the two query blocks: a statement that writes one element of a register tuple
//     counts, for hipcc's hazard recogniser, as a write of the whole tuple, and it pads an s_nop in front of a following
//     statement that touches any element of it;
//   * no statement reads the result of the one right before it (transcendental forwarding, v_permlane after VALU).
#define V_EXP2(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define V_PACK(w, lo, hi) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi))
#define V_SWAP32(a, b) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b))
#define V_MAX3I(r, a, b, c) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c))
#define V_MAX3A(r, b, c) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c))
#define V_MAX2A(r, b) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(b))
#define V_MAX2I(r, a, b) asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))

// ---- the softmax stream of one tile as a list of micro-operations (each one asm statement), in issue order
// VAR (bits): 1 = the K fragment rows are read in the order that makes the lane's eight scores of a k-step eight CONSECUTIVE keys, so the
// packed probabilities are the PV B operand as they stand (no v_permlane32_swap); 2 = row sums two scores at a time (v_pk_add_f32)
template <int SUMV, int VAR> struct Stream {
    static constexpr bool NOSWAP = (VAR & 1) != 0, PKSUM = SUMV && (VAR & 2);
    static constexpr int NADD = SUMV ? (PKSUM ? 8 : 16) : 0;      // row-sum micro-operations per chunk pair
    static constexpr int GRP = (NADD + 8) / 4;                    // one group = the adds of a word of BOTH query blocks, then its two packs
    static constexpr int PAIR = 16 + NADD + 8 + (NOSWAP ? 0 : 4); // one (key half, k-step) chunk of BOTH query blocks: 16 exp2, [adds,] 8 packs, [4 half swaps]
    static constexpr int EXP = 4 * PAIR;
    static constexpr int ALL = EXP + 32 + 2;       // + row maxima of the next tile (4 accumulators x 8, round-robin), 2 combines (this lane's half of the row)
    static constexpr int MID = 20;                 // issued as one burst behind the per-tile barrier (covers the V^T fragment latency)
    static constexpr int NGAP = SUMV ? 32 : 40;    // MFMAs per tile: 16 score, 16 PV [, 8 row-sum]
    static constexpr int PVG = SUMV ? 4 : 6;       // gaps per k-step unit in the PV phase
};
#define V_ACC(acc, x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x))
#define V_PKACC(acc, x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x))
typedef float f2v __attribute__((ext_vector_type(2)));

// One VALU micro-operation of the softmax stream.
// F < EXP, tile `cur`: chunk t = 2 kb + s holds this lane's scores of keys 16 t + 4 hi + {0..3} and 16 t + 8 + 4 hi + {0..3} (the
// accumulator layout); exp2 in place, (row sums,) packing to fp16 words w0..w3, then v_permlane32_swap(w0, w2), (w1, w3) with the
// lane that holds the other half of the query row: afterwards the lane owns P of keys 16 t + 8 hi + {0..7} - eight consecutive
// keys, the plain B-operand layout, so that the matching V^T fragment is ONE 16-byte chunk (ds_read_b128, no two-piece gather).
// Chunks come in the order the PV MFMAs consume them.  F >= EXP: row maxima over tile `nxt`.
// (separate scalars, not arrays, for the maxima and sums: an array becomes one register tuple, and a write to one element of a
// tuple makes hipcc pad an s_nop in front of the next statement that touches any other element)
template <int F, int SUMV, int VAR>
FI void filler(f16v (&cur)[2][2], f16v (&nxt)[2][2], u4v (&pf)[2][2][2], float& m00, float& m01, float& m10, float& m11, float& mx0, float& mx1,
               float& l00, float& l01, float& l10, float& l11, f2v& lp0, f2v& lp1) {
    using ST = Stream<SUMV, VAR>;
    if constexpr (F < ST::EXP) {
        constexpr int t = F / ST::PAIR, q = F % ST::PAIR, kb = t / 2, s = t % 2;
        if constexpr (q < 16) {
            constexpr int b = q % 2, e = q / 2;
            V_EXP2(cur[b][kb][8 * s + e]);
        } else if constexpr (q < 16 + ST::NADD + 8) {
            constexpr int idx = q - 16, grp = idx / ST::GRP, w = idx % ST::GRP, b = w % 2;
            if constexpr (w < ST::GRP - 2) {
                if constexpr (ST::PKSUM) {      // both scores of word grp at once: (l_b0, l_b1) += (S[2 grp], S[2 grp + 1])
                    const f2v pair = __builtin_shufflevector(cur[b][kb], cur[b][kb], 8 * s + 2 * grp, 8 * s + 2 * grp + 1);
                    if constexpr (b == 0) V_PKACC(lp0, pair); else V_PKACC(lp1, pair);
                } else {
                    constexpr int e = 2 * grp + w / 2;
                    float& ll = b == 0 ? (w / 2 == 0 ? l00 : l01) : (w / 2 == 0 ? l10 : l11);
                    V_ACC(ll, cur[b][kb][8 * s + e]);
                }
            } else {
                V_PACK(pf[b][kb][s][grp], cur[b][kb][8 * s + 2 * grp], cur[b][kb][8 * s + 2 * grp + 1]);
            }
        } else {
            constexpr int idx = q - (ST::PAIR - 4), b = idx % 2, which = idx / 2;
            V_SWAP32(pf[b][kb][s][which], pf[b][kb][s][which + 2]);
        }
    } else if constexpr (F < ST::EXP + 32) {
        constexpr int m = F - ST::EXP, o = m / 4, acc = m % 4, kb = acc / 2, b = acc % 2;       // the four chains interleaved
        float& mm = b == 0 ? (kb == 0 ? m00 : m01) : (kb == 0 ? m10 : m11);
        if constexpr (o == 0) V_MAX3I(mm, nxt[b][kb][0], nxt[b][kb][1], nxt[b][kb][2]);
        else if constexpr (o < 7) V_MAX3A(mm, nxt[b][kb][2 * o + 1], nxt[b][kb][2 * o + 2]);
        else V_MAX2A(mm, nxt[b][kb][15]);
    } else {
        if constexpr (F == ST::EXP + 32) V_MAX2I(mx0, m00, m01);
        else V_MAX2I(mx1, m10, m11);
    }
}
// The other 32 keys of a query row live in lane ^ 32: after the half swap t0 / t1 hold, in every lane, this lane's value and its
// partner's (in either order).  Not part of the hot stream: whether ANY row of the wave exceeds the deferred-max threshold is
// a ballot over the per-lane partial maxima; the exact row maximum is only needed once the (rare) rescale is taken.
FI float row_max_across_halves(float m) {
    float t0 = m, t1 = m, r;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(t0), "+v"(t1));
    V_MAX2I(r, t0, t1);
    return r;
}

// ABL: timing-only ablation bits (knob EXP0; 0 in production, anything else computes garbage): 1 no vmcnt wait at the barrier,
// 2 no barrier, 4 no exp2, 8 no softmax stream at all, 16 no fragment reads in the loop, 32 no DMA in the loop, 64 no MFMAs
// SUMV: row sums by v_add_f32 in the softmax stream (1) or by 8 extra MFMAs per tile (0)
template <int ABL, int SUMV, int VAR>
__global__ void __launch_bounds__(256, 1) flash2_d64_kernel(Flash2Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // four distinct objects: slot s of K / V^T (see the header: compile-time slots keep the DMA waits exact)
    __shared__ __attribute__((aligned(16))) half_t sK0[64 * 64];
    __shared__ __attribute__((aligned(16))) half_t sK1[64 * 64];
    __shared__ __attribute__((aligned(16))) half_t sV0[64 * 64];
    __shared__ __attribute__((aligned(16))) half_t sV1[64 * 64];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    // all query blocks of one (group, head) problem on ONE XCD (ids of the same residue mod 8), as flash_d64_kernel
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int prob = (slot / p.nqb) * 8 + xcd;
    if (prob >= p.nprob) return;
    const int g = prob / p.heads, h = prob % p.heads;
    const int q0 = ((slot % p.nqb) * 4 + wave) * 64;

    // ---- Q fragments (B operand): lane (q = lq, hi) holds Q[q][s*16 + hi*8 .. +7]; parked in a[64:95].  All eight loads are in
    // flight together (rows past nq read the last valid row: their results are never stored) and are only waited for behind the
    // first K / V^T DMA requests below.
    const half_t* qbase = p.q + ((int64_t)g * p.nq) * p.ldq + h * 64;
    bool qvalid[2];
    u4v qw[8];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qrow = q0 + b * 32 + lq;
        qvalid[b] = qrow < p.nq;
        const int qr = qvalid[b] ? qrow : p.nq - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qw[b * 4 + s] = *reinterpret_cast<const u4v*>(qbase + (int64_t)qr * p.ldq + s * 16 + hi * 8);
    }

    // ---- K / V^T streams
    const int64_t kvrow0 = (int64_t)(g / p.kv_div) * p.kv_rows;
    const half_t* kbase = p.k + kvrow0 * p.ldk + h * 64;
    const half_t* vbase = p.vt + (int64_t)(h * 64) * p.ldvt + kvrow0;
    const u32 k_bytes = (u32)(((int64_t)(p.nk - 1) * p.ldk + 64) * 2);
    const u32 v_bytes = (u32)((63ll * p.ldvt + p.nk) * 2);
    const __amdgpu_buffer_rsrc_t srd_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kbase), 0, (int)k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vbase), 0, (int)v_bytes, 0x00020000);
    // DMA map: 512 16-byte chunks per tile and operand, 2 per thread; source chunk swizzled, LDS image lane-linear
    const int srow = tid >> 3, spos = tid & 7;
    u32 koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = srow + 32 * i;
        const int csrc = spos ^ ((r >> 1) & 7);
        koff[i] = (u32)((int64_t)r * p.ldk * 2) + csrc * 16;
        voff[i] = (u32)((int64_t)r * p.ldvt * 2) + csrc * 16;
    }
    const u32 ktile_bytes = (u32)(64 * p.ldk * 2);        // K advances 64 rows per tile, V^T 64 columns = 128 bytes
    const int wslice = wave * 8 * 64;                     // this wave's 8 rows of each 32-row half of a slot
    auto dma_k = [&](auto slot_c, int kt) {
        half_t* dst = (decltype(slot_c)::value ? sK1 : sK0) + wslice;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)dst, 16, koff[0], (u32)kt * ktile_bytes, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(dst + 32 * 64), 16, koff[1], (u32)kt * ktile_bytes, 0, 0);
    };
    auto dma_v = [&](auto slot_c, int kt) {
        half_t* dst = (decltype(slot_c)::value ? sV1 : sV0) + wslice;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)dst, 16, voff[0], (u32)kt * 128u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(dst + 32 * 64), 16, voff[1], (u32)kt * 128u, 0, 0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;

    // ---- fragment addresses inside a slot (element offsets).  K fragment unit u = 4 kb + s: row kb*32 + lq, chunk 2 s + hi;
    // V^T fragment (unit t = 2 kb + s, d half db): row db*32 + lq, chunk 4 kb + 2 s + hi = 2 t + hi - the same four offsets
    // (the row swizzle depends on lq only; kb / db add 32 rows = an immediate)
    // VAR & 1: K fragment row i of a 32-key half holds key pi(i) = i with bits 2 and 3 exchanged.  The 32x32 accumulator puts row
    // i = 8 q + 4 hi + c into register 4 q + c of lane half hi, so registers 8 s .. 8 s + 7 of a lane then are keys 16 s + 8 hi + {0..7} -
    // eight consecutive keys, the PV B operand of k-step s as it stands (the rows of an MFMA's A operand are independent: any
    // assignment of keys to rows is legal, and the rows a 16-lane group reads stay the same 16 rows, conflict-free as before)
    int fa[4], fk[4];
    const int lqk = (VAR & 1) ? ((lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1)) : lq;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        fa[s] = tile_off2(lq, 2 * s + hi);
        fk[s] = tile_off2(lqk, 2 * s + hi);
    }
    h8 kf[4];                   // ring of K fragment units (slot u % 4), requested two units ahead
    h8 vf[3][2];                // ring of V^T fragment units (slot t % 3) x d half
    auto read_k = [&](const half_t* cK, auto u_c) {
        constexpr int u = decltype(u_c)::value;
        kf[u % 4] = *reinterpret_cast<const h8*>(cK + (u / 4) * 32 * 64 + fk[u % 4]);
    };
    auto read_v = [&](const half_t* cV, auto t_c) {
        constexpr int t = decltype(t_c)::value;
        vf[t % 3][0] = *reinterpret_cast<const h8*>(cV + fa[t]);
        vf[t % 3][1] = *reinterpret_cast<const h8*>(cV + 32 * 64 + fa[t]);
    };

    // ---- state (VGPRs)
    f16v S[2][2][2];            // [tile parity][query block][key half]: scores minus the running max (base-2 logits)
    f16v cinit[2];              // -running max of query block b in all 16 slots: C operand of a tile's first score MFMA
    u4v pf[2][2][2];            // packed fp16 probabilities [query block][key half][k-step] (8 halves = the B operand of a PV MFMA)
    float m00, m01, m10, m11, mx0, mx1;     // row maxima: partial (per score accumulator) and per query block
    float l00 = 0.f, l01 = 0.f, l10 = 0.f, l11 = 0.f;      // SUMV: row sums, two partial accumulators per query block
    f2v lp0 = {0.f, 0.f}, lp1 = {0.f, 0.f};                // ... as register pairs (VAR & 2)
    using ST = Stream<SUMV, VAR>;
    constexpr bool PKSUM = ST::PKSUM;
    constexpr int NF_EXP = ST::EXP, NF_ALL = ST::ALL, NF_MID = ST::MID, NF_PAIR = ST::PAIR, NGAP = ST::NGAP, PVG = ST::PVG;
    float negm[2] = {0.f, 0.f};

    const int ntiles = p.nk >> 6;

    // ---- prologue: K(0), V^T(0), K(1); S(0); K(2) into the slot S(0) has just finished with
    dma_k(C0{}, 0);
    dma_v(C0{}, 0);
    if (ntiles > 1) dma_k(C1{}, 1);
    sfor<0, 8>([&](auto i_c) { acc_load_q<decltype(i_c)::value>(qw[decltype(i_c)::value]); });
    acc_zero_o(0x3c003c00u);
    __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    sfor<0, 8>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value, kb = u / 4, s = u % 4;
        read_k(sK0, u_c);
        if constexpr (s == 0) {
            mfma_qk_zero<0>(S[0][0][kb], kf[u % 4]);
            mfma_qk_zero<4>(S[0][1][kb], kf[u % 4]);
        } else {
            mfma_qk_acc<s>(S[0][0][kb], kf[u % 4]);
            mfma_qk_acc<4 + s>(S[0][1][kb], kf[u % 4]);
        }
    });
    __builtin_amdgcn_s_barrier();                 // every wave has read K(0): its slot can take K(2)
    if (ntiles > 2) dma_k(C0{}, 2);
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(S[0][0][0]), "+v"(S[0][0][1]), "+v"(S[0][1][0]), "+v"(S[0][1][1]));   // MFMA results -> VALU
    // first tile: the max moves to the row maximum itself (nothing is accumulated yet)
    sfor<NF_EXP, NF_ALL>([&](auto f_c) { filler<decltype(f_c)::value, SUMV, VAR>(S[1], S[0], pf, m00, m01, m10, m11, mx0, mx1, l00, l01, l10, l11, lp0, lp1); });
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float m = row_max_across_halves(b ? mx1 : mx0);
        negm[b] = -m;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            S[0][b][0][i] -= m;
            S[0][b][1][i] -= m;
            cinit[b][i] = negm[b];
        }
    }
    read_k(sK1, C0{});                            // K fragment units 0, 1 of tile 1 (garbage if there is none: never used)
    read_k(sK1, C1{});

    // ---- one tile: PAR = parity of tile kt (compile time), HAS_NEXT = tile kt + 1 exists.  40 MFMA gaps:
    //   0-15   score MFMAs of tile kt + 1: gap 2 u + b, K fragment unit u = 4 kb + s, query block b
    //   16-39  per k-step unit t = 2 kb + s of tile kt six gaps: PV (db, b) x 4, then the two row-sum MFMAs
    // LDS fragments are requested in pairs of units four or more MFMAs ahead and waited for with ONE explicit lgkmcnt(0) in front
    // of the first use (the compiler's own counted waits - one per fragment - then vanish: an s_waitcnt is an issue slot too).
    auto step = [&](auto par_c, auto next_c, int kt) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool HAS_NEXT = decltype(next_c)::value != 0;
        constexpr int NF = HAS_NEXT ? NF_ALL : NF_EXP;
        const half_t* cK = (PAR ^ 1) ? sK1 : sK0;          // K(kt + 1)
        const half_t* cV = PAR ? sV1 : sV0;                // V^T(kt)
        const half_t* nK = PAR ? sK1 : sK0;                // K(kt + 2)
        sfor<0, NGAP>([&](auto g_c) {
            constexpr int gp = decltype(g_c)::value;
            // (0) fragments due in this group of gaps have arrived; request the next pair
            if constexpr (HAS_NEXT && gp < 16 && gp % 4 == 0) {
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0)
                if constexpr (gp < 12 && !(ABL & 16)) {
                    read_k(cK, std::integral_constant<int, gp / 2 + 2>{});
                    read_k(cK, std::integral_constant<int, gp / 2 + 3>{});
                }
            }
            if constexpr (gp == 16 + 3 * PVG) {
                __builtin_amdgcn_s_waitcnt(0xc07f);        // V^T unit 3 (requested one unit after unit 0's last PV MFMA)
                if constexpr (HAS_NEXT && !(ABL & 16)) {   // K fragment units 0, 1 of tile kt + 2: five MFMAs ahead of the next step's gap 0
                    read_k(nK, C0{});
                    read_k(nK, C1{});
                }
            }
            // (1) the MFMA of this gap
            if constexpr (ABL & 64) {
            } else if constexpr (gp < 16) {
                if constexpr (HAS_NEXT) {
                    constexpr int u = gp / 2, b = gp % 2, kb = u / 4, s = u % 4;
                    if constexpr (s == 0) mfma_qk_first<4 * b>(S[PAR ^ 1][b][kb], kf[u % 4], cinit[b]);
                    else mfma_qk_acc<4 * b + s>(S[PAR ^ 1][b][kb], kf[u % 4]);
                }
            } else {
                constexpr int t = (gp - 16) / PVG, r = (gp - 16) % PVG, kb = t / 2, s = t % 2;
                if constexpr (r < 4) mfma_pv<2 * (r % 2) + r / 2>(vf[t % 3][r / 2], pf[r % 2][kb][s]);
                else mfma_rowsum<r - 4>(pf[r - 4][kb][s]);          // (PVG = 6 only)
            }
            // (2) its share of the softmax stream (NF_MID of it runs as a burst behind the barrier, see (4))
            constexpr int NFG = NF - NF_MID;
            constexpr int f0 = HAS_NEXT ? gp * NFG / NGAP + (gp >= 16 ? NF_MID : 0) : (gp < 16 ? gp * (NF_EXP / 16) : NF);
            constexpr int f1 = HAS_NEXT ? (gp + 1) * NFG / NGAP + (gp >= 15 ? NF_MID : 0) : (gp < 16 ? (gp + 1) * (NF_EXP / 16) : NF);
            constexpr int fmid = HAS_NEXT ? f1 - NF_MID : f1;       // gap 15: [f0, fmid) before the barrier, [fmid, f1) behind it
            sfor<f0, (gp == 15 ? fmid : f1)>([&](auto f_c) {
                constexpr int F = decltype(f_c)::value;
                if constexpr (!(ABL & 8) && !((ABL & 4) && F < NF_EXP && F % NF_PAIR < 16)) filler<F, SUMV, VAR>(S[PAR], S[PAR ^ 1], pf, m00, m01, m10, m11, mx0, mx1, l00, l01, l10, l11, lp0, lp1);
            });
            // (3) later requests
            if constexpr (gp == 16 + PVG && !(ABL & 16)) read_v(cV, std::integral_constant<int, 3>{});      // slot 0: unit 0's PV MFMAs are gaps 16-19
            if constexpr (HAS_NEXT && gp == 18 && !(ABL & 32)) dma_v(std::integral_constant<int, PAR ^ 1>{}, kt + 1);
            // (4) between the two MFMA phases: everything this wave has in flight by DMA has landed, then all waves meet.  Only now
            // is V^T(kt) (issued one tile ago) complete for EVERY wave, so its fragments are requested here and a burst of the
            // softmax stream covers their LDS latency; K(kt + 1) is no longer read by anybody and V^T(kt - 1) neither: their
            // slots take K(kt + 3) (here) and V^T(kt + 1) (three gaps on, beside the MFMAs)
            if constexpr (gp == 15) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(ABL & 1)) __builtin_amdgcn_s_waitcnt(0x0f70);
                if constexpr (!(ABL & 2)) __builtin_amdgcn_s_barrier();
                if constexpr (!(ABL & 16)) {
                    read_v(cV, C0{});
                    read_v(cV, C1{});
                    read_v(cV, std::integral_constant<int, 2>{});
                }
                if constexpr (HAS_NEXT && !(ABL & 32)) {
                    if (kt + 3 < ntiles) dma_k(std::integral_constant<int, PAR ^ 1>{}, kt + 3);
                }
                __builtin_amdgcn_sched_barrier(0);
                sfor<fmid, f1>([&](auto f_c) {
                    constexpr int F = decltype(f_c)::value;
                    if constexpr (!(ABL & 8) && !((ABL & 4) && F < NF_EXP && F % NF_PAIR < 16)) filler<F, SUMV, VAR>(S[PAR], S[PAR ^ 1], pf, m00, m01, m10, m11, mx0, mx1, l00, l01, l10, l11, lp0, lp1);
                });
                __builtin_amdgcn_s_waitcnt(0xc07f);        // V^T units 0-2
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // (5) does the running max have to move for tile kt + 1?  (deferred: only when a row would exceed it by more than 2^8)
        if constexpr (HAS_NEXT) {
            if (__builtin_amdgcn_ballot_w64(fmaxf(mx0, mx1) > F2_DEFER) != 0) {       // wave-uniform, rare
                float delta[2], alpha[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    delta[b] = fmaxf(row_max_across_halves(b ? mx1 : mx0), 0.f);
                    alpha[b] = __builtin_amdgcn_exp2f(-delta[b]);
                    negm[b] -= delta[b];
                    if constexpr (PKSUM) {
                        if (b == 0) { lp0[0] *= alpha[0]; lp0[1] *= alpha[0]; }
                        else { lp1[0] *= alpha[1]; lp1[1] *= alpha[1]; }
                    } else if constexpr (SUMV) {
                        if (b == 0) { l00 *= alpha[0]; l01 *= alpha[0]; }
                        else { l10 *= alpha[1]; l11 *= alpha[1]; }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        S[PAR ^ 1][b][0][i] -= delta[b];
                        S[PAR ^ 1][b][1][i] -= delta[b];
                        cinit[b][i] = negm[b];
                    }
                }
                acc_scale_o<0>(alpha[0]);
                acc_scale_o<1>(alpha[1]);       // (ends in s_nop 1: the C tuples above are VALU-written MFMA sources)
            }
        }
    };

    int kt = 0;
    for (; kt + 2 < ntiles; kt += 2) {
        step(C0{}, C1{}, kt);
        step(C1{}, C1{}, kt + 1);
    }
    if (kt + 2 == ntiles) {
        step(C0{}, C1{}, kt);
        step(C1{}, C0{}, kt + 1);
    } else {
        step(C0{}, C0{}, kt);
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l
    sfor<0, 2>([&](auto b_c) {
        constexpr int b = decltype(b_c)::value;
        float l_tot;
        if constexpr (SUMV) {
            const float lsum = PKSUM ? (b == 0 ? lp0[0] + lp0[1] : lp1[0] + lp1[1]) : (b == 0 ? l00 + l01 : l10 + l11);
            l_tot = lsum + __shfl_xor(lsum, 32);
        } else {
            l_tot = acc_read_l<b>();                       // both key halves of the row are in the MFMA row sum already
        }
        const float inv = 1.0f / l_tot;
        f16v o[2];
        acc_read_o<2 * b>(o[0]);
        acc_read_o<2 * b + 1>(o[1]);
        if (qvalid[b]) {
            half_t* orow = p.o + ((int64_t)g * p.nq + q0 + b * 32 + lq) * p.ldo + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int d0 = db * 32 + 8 * gq + 4 * hi;
                    *reinterpret_cast<h4*>(orow + d0) = h4{(half_t)(o[db][gq * 4 + 0] * inv), (half_t)(o[db][gq * 4 + 1] * inv),
                                                           (half_t)(o[db][gq * 4 + 2] * inv), (half_t)(o[db][gq * 4 + 3] * inv)};
                }
        }
    });
#endif
}

}  // namespace

int vcx_flash2_launch(const Flash2Args& a, hipStream_t s) {
    const int prob_pad = (a.nprob + 7) / 8 * 8;
    const dim3 grid((unsigned)(a.nqb * prob_pad));
#define F2_LAUNCH(A, SV) hipLaunchKernelGGL((flash2_d64_kernel<A, SV, F2_VAR_DEFAULT>), grid, dim3(256), 0, s, a)
#define F2_LAUNCHV(A, SV, V) hipLaunchKernelGGL((flash2_d64_kernel<A, SV, V>), grid, dim3(256), 0, s, a)
#ifndef VCX_FLASH2_ABLATIONS
    // the product: one kernel, no knob read - the scratch knobs EXP0 / EXP1 ("free for one-off experiments", vcx.h) must not be
    // able to turn every 9216-key self-attention into VCX_EINVAL because some unrelated experiment set them
    F2_LAUNCH(0, F2_SUMV_DEFAULT);
#else
    const int abl = vcx_tune(VCX_TUNE_EXP0), sumv = vcx_tune(VCX_TUNE_EXP1);      // timing-only ablations / A-B (tools/flash_ab.py); 0, 0 = the product
    if (abl == 0 && sumv == 0) F2_LAUNCH(0, F2_SUMV_DEFAULT);
    else if (abl == 0 && sumv == 1) F2_LAUNCH(0, 1);
    else if (abl == 0 && sumv == 2) F2_LAUNCH(0, 0);
    else if (abl == 0 && sumv >= 10 && sumv <= 13) {       // A/B of the stream variants: EXP1 = 10 + VAR
        if (sumv == 10) F2_LAUNCHV(0, 1, 0);
        else if (sumv == 11) F2_LAUNCHV(0, 1, 1);
        else if (sumv == 12) F2_LAUNCHV(0, 1, 2);
        else F2_LAUNCHV(0, 1, 3);
    }
    else if (abl == 1) F2_LAUNCH(1, F2_SUMV_DEFAULT);
    else if (abl == 3) F2_LAUNCH(3, F2_SUMV_DEFAULT);
    else if (abl == 4) F2_LAUNCH(4, F2_SUMV_DEFAULT);
    else if (abl == 8) F2_LAUNCH(8, F2_SUMV_DEFAULT);
    else if (abl == 16) F2_LAUNCH(16, F2_SUMV_DEFAULT);
    else if (abl == 32) F2_LAUNCH(32, F2_SUMV_DEFAULT);
    else if (abl == 35) F2_LAUNCH(35, F2_SUMV_DEFAULT);
    else if (abl == 64) F2_LAUNCH(64, F2_SUMV_DEFAULT);
    else if (abl == 72) F2_LAUNCH(72, F2_SUMV_DEFAULT);
    else if (abl == 59) F2_LAUNCH(59, F2_SUMV_DEFAULT);
    else {
        vcx_set_error("vcx_attn_flash_d64_f16(v2): ablation variant EXP0=%d EXP1=%d does not exist", abl, sumv);
        return VCX_EINVAL;
    }
#endif
#undef F2_LAUNCH
#undef F2_LAUNCHV
    return vcx_check_launch("vcx_attn_flash_d64_f16(v2)");
}
