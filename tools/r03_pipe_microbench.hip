// r03_pipe_microbench.hip -- round-3 recalibration of what the gfx950 CU charges for the instruction classes pv_wave_kernel_1024 is made of.
//
// The round-2 tables (tools/valu_microbench*.hip) reported "cycles" from s_memtime alone and read v_add_f32 at 1.73 cycles per SIMD, below the
// 2-cycle wave64 issue of a SIMD-32: the unit was off.  Here every row is timed THREE ways at once -- s_memtime (shader-clock ticks),
// s_memrealtime (100 MHz constant clock) and host HIP events around the launch -- and the tick is calibrated against the one instruction whose
// cost is known by construction: a saturated SIMD retires one wave64 v_add_f32 per 2 shader cycles (MI355X_MICROARCH.md, Wave scheduling).
//
// One workgroup per CU (96 KB of dynamic LDS keeps a second one off), W waves, every wave runs ITER iterations of a body of NB instructions.
// Reported per row: ns per wave-instruction per CU-pipe (LDS rows) or per SIMD (VALU rows), in ticks, and in calibrated cycles.
//   build: hipcc -O3 --offload-arch=gfx950 -o r03_pipe_microbench r03_pipe_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum Body {
    V_ADD_F32, V_FMA_F32, V_FMA_F64, V_ADD_F64, V_MUL_F64, V_PK_FMA_F32, V_PK_ADD_F32, V_MAX3_U32, V_CNDMASK, V_PERM, V_PERMLANE32_SWAP, V_PERMLANE16_SWAP,
    V_MOV_DPP_ROR8, V_MOV_DPP_QUAD, V_CVT_F32_F64, V_CVT_F64_F32,
    DS_W128, DS_W64, DS_W32, DS_WADDTID, DS_R128, DS_R64, DS_R32, DS_BPERM,
    XCH_F64,      // 8 ds_write_b128, fence, 8 ds_read_b128, wait: one fp64 transpose of the forward FFT
    XCH_F32,      // 4 ds_write_b128, 8 ds_read_b64, wait: one packed-fp32 transpose of the inverse FFT
    FRAME_LDS,    // the LDS instruction sequence of ONE f >= 1 frame of pv_wave_kernel_1024<2>, no arithmetic
    FRAME_MIX,    // the same with the frame's VALU instruction mix between the exchanges (301 fp64, 154 packed, ~320 plain)
    NBODIES
};
static const char *body_name[] = {"v_add_f32", "v_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_pk_fma_f32", "v_pk_add_f32", "v_max3_u32", "v_cndmask_b32", "v_perm_b32",
                                  "v_permlane32_swap", "v_permlane16_swap", "v_mov_b32 dpp row_ror:8", "v_mov_b32 dpp quad_perm", "v_cvt_f32_f64", "v_cvt_f64_f32",
                                  "ds_write_b128", "ds_write_b64", "ds_write_b32", "ds_write_addtid_b32", "ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_bpermute_b32",
                                  "exchange fp64 (8 w128 + 8 r128)", "exchange fp32 (4 w128 + 8 r64)", "frame: LDS sequence only", "frame: LDS sequence + VALU mix"};

#define REP8(x) x x x x x x x x
#define REP4(x) x x x x

struct Regs {
    double d[8], dc;
    v2f p[8], pc;
    float f[8], fc;
    unsigned u[8], uc;
    v4f lv;
};

template <int B> __device__ __forceinline__ void valu_block(Regs &r)
{
    // 8 independent instructions of class B
    if (B == V_ADD_F32) asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8" : "+v"(r.f[0]), "+v"(r.f[1]), "+v"(r.f[2]), "+v"(r.f[3]), "+v"(r.f[4]), "+v"(r.f[5]), "+v"(r.f[6]), "+v"(r.f[7]) : "v"(r.fc));
    if (B == V_FMA_F32) asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\nv_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8" : "+v"(r.f[0]), "+v"(r.f[1]), "+v"(r.f[2]), "+v"(r.f[3]), "+v"(r.f[4]), "+v"(r.f[5]), "+v"(r.f[6]), "+v"(r.f[7]) : "v"(r.fc));
    if (B == V_FMA_F64) asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\nv_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8" : "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]), "+v"(r.d[4]), "+v"(r.d[5]), "+v"(r.d[6]), "+v"(r.d[7]) : "v"(r.dc));
    if (B == V_ADD_F64) asm volatile("v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\nv_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8" : "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]), "+v"(r.d[4]), "+v"(r.d[5]), "+v"(r.d[6]), "+v"(r.d[7]) : "v"(r.dc));
    if (B == V_MUL_F64) asm volatile("v_mul_f64 %0, %0, %8\nv_mul_f64 %1, %1, %8\nv_mul_f64 %2, %2, %8\nv_mul_f64 %3, %3, %8\nv_mul_f64 %4, %4, %8\nv_mul_f64 %5, %5, %8\nv_mul_f64 %6, %6, %8\nv_mul_f64 %7, %7, %8" : "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]), "+v"(r.d[4]), "+v"(r.d[5]), "+v"(r.d[6]), "+v"(r.d[7]) : "v"(r.dc));
    if (B == V_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_pk_fma_f32 %1, %1, %8, %8\nv_pk_fma_f32 %2, %2, %8, %8\nv_pk_fma_f32 %3, %3, %8, %8\nv_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\nv_pk_fma_f32 %6, %6, %8, %8\nv_pk_fma_f32 %7, %7, %8, %8" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]), "+v"(r.p[4]), "+v"(r.p[5]), "+v"(r.p[6]), "+v"(r.p[7]) : "v"(r.pc));
    if (B == V_PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\nv_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]), "+v"(r.p[4]), "+v"(r.p[5]), "+v"(r.p[6]), "+v"(r.p[7]) : "v"(r.pc));
    if (B == V_MAX3_U32) asm volatile("v_max3_u32 %0, %0, %8, %1\nv_max3_u32 %1, %1, %8, %2\nv_max3_u32 %2, %2, %8, %3\nv_max3_u32 %3, %3, %8, %4\nv_max3_u32 %4, %4, %8, %5\nv_max3_u32 %5, %5, %8, %6\nv_max3_u32 %6, %6, %8, %7\nv_max3_u32 %7, %7, %8, %0" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]), "+v"(r.u[4]), "+v"(r.u[5]), "+v"(r.u[6]), "+v"(r.u[7]) : "v"(r.uc));
    if (B == V_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]), "+v"(r.u[4]), "+v"(r.u[5]), "+v"(r.u[6]), "+v"(r.u[7]) : "v"(r.uc) : "vcc");
    if (B == V_PERM) asm volatile("v_perm_b32 %0, %0, %8, %8\nv_perm_b32 %1, %1, %8, %8\nv_perm_b32 %2, %2, %8, %8\nv_perm_b32 %3, %3, %8, %8\nv_perm_b32 %4, %4, %8, %8\nv_perm_b32 %5, %5, %8, %8\nv_perm_b32 %6, %6, %8, %8\nv_perm_b32 %7, %7, %8, %8" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]), "+v"(r.u[4]), "+v"(r.u[5]), "+v"(r.u[6]), "+v"(r.u[7]) : "v"(r.uc));
    if (B == V_PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\nv_permlane32_swap_b32 %0, %2\nv_permlane32_swap_b32 %1, %3\nv_permlane32_swap_b32 %4, %6\nv_permlane32_swap_b32 %5, %7" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]), "+v"(r.u[4]), "+v"(r.u[5]), "+v"(r.u[6]), "+v"(r.u[7]));
    if (B == V_PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\nv_permlane16_swap_b32 %6, %7\nv_permlane16_swap_b32 %0, %2\nv_permlane16_swap_b32 %1, %3\nv_permlane16_swap_b32 %4, %6\nv_permlane16_swap_b32 %5, %7" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]), "+v"(r.u[4]), "+v"(r.u[5]), "+v"(r.u[6]), "+v"(r.u[7]));
    if (B == V_MOV_DPP_ROR8) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc\nv_mov_b32_dpp %1, %2 row_ror:8 row_mask:0xf bank_mask:0x3\nv_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xc\nv_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0x3\nv_mov_b32_dpp %4, %5 row_ror:8 row_mask:0xf bank_mask:0xc\nv_mov_b32_dpp %5, %6 row_ror:8 row_mask:0xf bank_mask:0x3\nv_mov_b32_dpp %6, %7 row_ror:8 row_mask:0xf bank_mask:0xc\nv_mov_b32_dpp %7, %0 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]), "+v"(r.u[4]), "+v"(r.u[5]), "+v"(r.u[6]), "+v"(r.u[7]));
    if (B == V_MOV_DPP_QUAD) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]), "+v"(r.u[4]), "+v"(r.u[5]), "+v"(r.u[6]), "+v"(r.u[7]));
    if (B == V_CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %8\nv_cvt_f32_f64 %2, %8\nv_cvt_f32_f64 %3, %8\nv_cvt_f32_f64 %4, %8\nv_cvt_f32_f64 %5, %8\nv_cvt_f32_f64 %6, %8\nv_cvt_f32_f64 %7, %8" : "+v"(r.f[0]), "+v"(r.f[1]), "+v"(r.f[2]), "+v"(r.f[3]), "+v"(r.f[4]), "+v"(r.f[5]), "+v"(r.f[6]), "+v"(r.f[7]) : "v"(r.dc));
    if (B == V_CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %8\nv_cvt_f64_f32 %2, %8\nv_cvt_f64_f32 %3, %8\nv_cvt_f64_f32 %4, %8\nv_cvt_f64_f32 %5, %8\nv_cvt_f64_f32 %6, %8\nv_cvt_f64_f32 %7, %8" : "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]), "+v"(r.d[4]), "+v"(r.d[5]), "+v"(r.d[6]), "+v"(r.d[7]) : "v"(r.fc));
}

// LDS primitives on the wave's private 9 KB region (a = byte address of this lane's 16-byte slot, row pitch 1152 B as in the kernel)
#define W128(off) asm volatile("ds_write_b128 %0, %1 offset:" #off ::"v"(a), "v"(r.lv) : "memory");
#define W64(off) asm volatile("ds_write_b64 %0, %1 offset:" #off ::"v"(a8), "v"(r.p[0]) : "memory");
#define W32(off) asm volatile("ds_write_b32 %0, %1 offset:" #off ::"v"(a4), "v"(r.f[0]) : "memory");
#define WTID(off) asm volatile("ds_write_addtid_b32 %0 offset:" #off ::"v"(r.f[0]) : "memory");
#define R128(off) { v4f x; asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(x) : "v"(a)); sink4(x); }
#define R64(off) { v2f x; asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(x) : "v"(a8)); sink2(x); }
#define R32(off) { float x; asm volatile("ds_read_b32 %0, %1 offset:" #off : "=v"(x) : "v"(a4)); sink1(x); }
#define WAITL asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
__device__ __forceinline__ void sink4(v4f x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void sink2(v2f x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void sink1(float x) { asm volatile("" ::"v"(x)); }

template <int NV64, int NPK, int NPL> __device__ __forceinline__ void valu_mix(Regs &r)
{
    // NV64 fp64 + NPK packed + NPL plain instructions in blocks of 8 (rounded down)
#pragma unroll
    for (int i = 0; i < NV64 / 8; i++) valu_block<V_FMA_F64>(r);
#pragma unroll
    for (int i = 0; i < NPK / 8; i++) valu_block<V_PK_FMA_F32>(r);
#pragma unroll
    for (int i = 0; i < NPL / 8; i++) valu_block<V_ADD_F32>(r);
}

// One f >= 1 frame of pv_wave_kernel_1024<2>: LDS instruction sequence in program order, with the waits where the kernel consumes data.
// MIX adds the VALU instruction mix of the phase in front of each exchange (counts from the ISA of the product build, rounded to blocks of 8).
template <bool MIX> __device__ __forceinline__ void frame_body(Regs &r, unsigned a, unsigned a8, unsigned a4)
{
    // forward pass 1: 7 twiddle reads (b128), radix-8 + 7 cmul in fp64
    R128(0) R128(1024) R128(2048) R128(3072) R128(4096) R128(5120) R128(6144) WAITL
    if (MIX) valu_mix<96, 0, 16>(r);
    W128(0) W128(1152) W128(2304) W128(3456) W128(4608) W128(5760) W128(6912) W128(8064)
    R128(0) R128(1152) R128(2304) R128(3456) R128(4608) R128(5760) R128(6912) R128(8064)
    R128(64) R128(192) R128(320) R128(448) R128(576) R128(704) R128(832) WAITL                 // second twiddle table
    if (MIX) valu_mix<96, 0, 8>(r);
    W128(0) W128(1152) W128(2304) W128(3456) W128(4608) W128(5760) W128(6912) W128(8064)
    R128(0) R128(1152) R128(2304) R128(3456) R128(4608) R128(5760) R128(6912) R128(8064) WAITL
    if (MIX) valu_mix<64, 0, 8>(r);                                                            // pass 3
    // split pass: partner exchange (4 w128 + 4 r128), |X|^2 (8 w32)
    W128(0) W128(1024) W128(2048) W128(3072)
    R128(16) R128(1040) R128(2064) R128(3088) WAITL
    if (MIX) valu_mix<48, 0, 40>(r);
    W32(4112) W32(4368) W32(4624) W32(4880) W32(5136) W32(5392) W32(5648) W32(5904)
    // peak search: 2 r64 + 2 r128 of magnitudes, 1 r128 of the shift table, 3 bpermutes, 2 w128 of routes
    R64(4112) R128(4128) R128(4144) R64(4160) R128(10336) WAITL
    if (MIX) valu_mix<0, 0, 96>(r);
    { unsigned x; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(x) : "v"(a4), "v"(r.u[0])); asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(x) : "v"(a4), "v"(r.u[1])); asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(x) : "v"(a4), "v"(r.u[2])); asm volatile("" ::"v"(x)); }
    WAITL
    if (MIX) valu_mix<0, 0, 56>(r);
    W128(4112) W128(5136)
    // zero Y (4 w128), scatter: 9 route reads (b32) + 9 Y writes (b64)
    W128(0) W128(1024) W128(2048) W128(3072)
    R32(4112) R32(4368) R32(4624) R32(4880) R32(5136) R32(5392) R32(5648) R32(5904) R32(6160) WAITL
    if (MIX) valu_mix<0, 0, 32>(r);
    W64(0) W64(512) W64(1024) W64(1536) W64(2048) W64(2560) W64(3072) W64(3584) W64(4096)
    // c2r pre-pass: 9 r64 of Y, hand-over 4 w64 + 4 r64
    R64(0) R64(512) R64(1024) R64(1536) R64(2048) R64(2560) R64(3072) R64(3584) R64(4096) WAITL
    if (MIX) valu_mix<0, 32, 8>(r);
    W64(6224) W64(6736) W64(7248) W64(7760)
    R64(6232) R64(6744) R64(7256) R64(7768) WAITL
    // inverse pass 1: 4 twiddle r128, radix-8 + 7 cmul packed
    R128(0) R128(1024) R128(2048) R128(3072) WAITL
    if (MIX) valu_mix<0, 40, 0>(r);
    W128(0) W128(1152) W128(2304) W128(3456)
    R64(0) R64(128) R64(256) R64(384) R64(512) R64(640) R64(768) R64(896)
    R128(64) R128(192) R128(320) R128(448) WAITL
    if (MIX) valu_mix<0, 40, 0>(r);
    W128(0) W128(1152) W128(2304) W128(3456)
    R64(0) R64(128) R64(256) R64(384) R64(512) R64(640) R64(768) R64(896)
    R128(8192) R128(9216) R128(10240) R128(11264) WAITL                                       // Hann rows
    if (MIX) valu_mix<0, 40, 56>(r);                                                          // pass 3, window, overlap-add
}

template <int B>
__global__ __launch_bounds__(1024) void kern(unsigned long long *out, int iters, float seed, unsigned simd_mask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 24576; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 0.f;
    __syncthreads();
    Regs r;
    for (int i = 0; i < 8; i++) { r.d[i] = seed + i; r.p[i] = v2f{seed, (float)i}; r.f[i] = seed + i; r.u[i] = l + i; }
    r.dc = 1.0 + 1e-9 * seed; r.pc = v2f{1.0001f, 0.9999f}; r.fc = 1.0001f; r.uc = 0x05040100u; r.lv = v4f{1.f, 2.f, 3.f, 4.f};
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    const unsigned simd = (hwid >> 4) & 3u;
    const bool active = (simd_mask >> simd) & 1u;
    // private 12 KB per wave; lane slot of 16 / 8 / 4 bytes
    const unsigned base = (unsigned)(wv & 7) * 12288u;
    const unsigned a = base + (unsigned)l * 16u, a8 = base + (unsigned)l * 8u, a4 = base + (unsigned)l * 4u;
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(base)));
    (void)a; (void)a8; (void)a4;
    __syncthreads();
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    if (active) {
        for (int i = 0; i < iters; i++) {
            if (B < DS_W128) { REP4(valu_block<B>(r);) }
            else if (B == DS_W128) { W128(0) W128(1152) W128(2304) W128(3456) W128(4608) W128(5760) W128(6912) W128(8064) WAITL }
            else if (B == DS_W64) { W64(0) W64(576) W64(1152) W64(1728) W64(2304) W64(2880) W64(3456) W64(4032) WAITL }
            else if (B == DS_W32) { W32(0) W32(288) W32(576) W32(864) W32(1152) W32(1440) W32(1728) W32(2016) WAITL }
            else if (B == DS_WADDTID) { WTID(0) WTID(256) WTID(512) WTID(768) WTID(1024) WTID(1280) WTID(1536) WTID(1792) WAITL }
            else if (B == DS_R128) { R128(0) R128(1152) R128(2304) R128(3456) R128(4608) R128(5760) R128(6912) R128(8064) WAITL }
            else if (B == DS_R64) { R64(0) R64(576) R64(1152) R64(1728) R64(2304) R64(2880) R64(3456) R64(4032) WAITL }
            else if (B == DS_R32) { R32(0) R32(288) R32(576) R32(864) R32(1152) R32(1440) R32(1728) R32(2016) WAITL }
            else if (B == DS_BPERM) {
                unsigned x = r.u[0];
                REP8(asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(x) : "v"(a4));)
                WAITL
                r.u[0] = x;
            } else if (B == XCH_F64) {
                W128(0) W128(1152) W128(2304) W128(3456) W128(4608) W128(5760) W128(6912) W128(8064)
                R128(0) R128(1152) R128(2304) R128(3456) R128(4608) R128(5760) R128(6912) R128(8064) WAITL
            } else if (B == XCH_F32) {
                W128(0) W128(1152) W128(2304) W128(3456)
                R64(0) R64(128) R64(256) R64(384) R64(512) R64(640) R64(768) R64(896) WAITL
            } else if (B == FRAME_LDS) frame_body<false>(r, a, a8, a4);
            else if (B == FRAME_MIX) frame_body<true>(r, a, a8, a4);
        }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    if (l == 0) {
        unsigned long long *o = out + 4 * (blockIdx.x * (blockDim.x >> 6) + wv);
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = active; o[3] = simd;
    }
    double acc = 0; for (int i = 0; i < 8; i++) acc += r.d[i] + r.p[i].x + r.f[i] + r.u[i];
    if (acc == 12345.678) out[0] = 0;
}

struct Res { double ticks, rt_ns, ev_ns; int nactive; };

template <int B> static Res run(int waves, int iters, unsigned simd_mask)
{
    const int blocks = 256;
    unsigned long long *c;
    hipMalloc(&c, sizeof(unsigned long long) * 4 * blocks * waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern<B>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL((kern<B>), dim3(blocks), dim3(64 * waves), 98304, 0, c, iters, 1.5f, simd_mask);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kern<B>), dim3(blocks), dim3(64 * waves), 98304, 0, c, iters, 1.5f, simd_mask);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(4 * blocks * waves);
    hipMemcpy(h.data(), c, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    hipFree(c); hipEventDestroy(e0); hipEventDestroy(e1);
    double st = 0, sr = 0; int n = 0;
    for (int i = 0; i < blocks * waves; i++) if (h[4 * i + 2]) { st += (double)h[4 * i]; sr += (double)h[4 * i + 1]; n++; }
    Res r; r.ticks = st / n; r.rt_ns = sr / n * 10.0; r.ev_ns = ms * 1e6; r.nactive = n / blocks;
    return r;
}

static double g_tick_per_cycle = 1.0;   // calibrated: ticks per shader cycle (v_add_f32 saturated = 2 cycles per SIMD)

template <int B> static void row(const char *unit_note, int instr_per_iter, bool per_simd, unsigned simd_mask = 0xF, const int *wlist = nullptr, int nw = 0)
{
    static const int dflt[] = {4, 8, 12, 16};
    if (!wlist) { wlist = dflt; nw = 4; }
    printf("%-34s", body_name[B]);
    for (int wi = 0; wi < nw; wi++) {
        const int w = wlist[wi];
        const int iters = (B >= FRAME_LDS) ? 300 : 2000;
        const Res r = run<B>(w, iters, simd_mask);
        const double n = (double)iters * instr_per_iter;
        // pipe occupancy per wave-instruction: per SIMD = wave time / n / (active waves per SIMD); per CU pipe = wave time / n / active waves
        const double share = per_simd ? (double)r.nactive / __builtin_popcount(simd_mask & 0xF) : (double)r.nactive;
        const double ticks = r.ticks / n / share, ns = r.rt_ns / n / share;
        printf(" | w=%2d: %6.2f tk %6.2f cyc %6.3f ns [ev %6.3f] (%4.0f MHz)", r.nactive, ticks, ticks / g_tick_per_cycle, ns, r.ev_ns / n / share, r.ticks / r.rt_ns * 1e3);
    }
    printf("  %s\n", unit_note);
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, hipDeviceAttributeClockRate %d kHz, hipDeviceAttributeWallClockRate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, clk, wclk);
    for (int it : {1000, 4000, 16000, 64000}) {   // is the gap between the in-kernel clocks and the HIP events a constant (launch overhead) or a factor (clock rate)?
        const Res r = run<V_ADD_F32>(16, it, 0xF);
        printf("v_add_f32 x %d iterations, 16 waves/CU: in-kernel %.1f us by s_memrealtime (at 100 MHz), %.1f us by s_memtime at 2.4 GHz, HIP events %.1f us\n",
               it, r.rt_ns * 1e-3, r.ticks / 2400.0, r.ev_ns * 1e-3);
    }
    {   // calibration: saturated SIMD (4 waves per SIMD), v_add_f32 = 2 shader cycles per wave-instruction
        const Res r = run<V_ADD_F32>(16, 4000, 0xF);
        const double ticks_per_instr_simd = r.ticks / (4000.0 * 32) / 4.0;
        g_tick_per_cycle = ticks_per_instr_simd / 2.0;
        printf("calibration: v_add_f32, 16 waves/CU: %.3f s_memtime ticks and %.3f ns per wave-instruction per SIMD; events %.3f ns; s_memtime runs at %.0f MHz against the 100 MHz clock\n",
               ticks_per_instr_simd, r.rt_ns / (4000.0 * 32) / 4.0, r.ev_ns / (4000.0 * 32) / 4.0, r.ticks / r.rt_ns * 1e3);
        printf("=> if the SIMD retires one wave64 v_add_f32 per 2 shader cycles, one s_memtime tick = %.3f shader cycles and the shader clock under this load is %.0f MHz\n",
               1.0 / g_tick_per_cycle, 2.0 / (r.rt_ns / (4000.0 * 32) / 4.0) * 1e3);
    }
    printf("\n== VALU: cost per wave-instruction PER SIMD (tk = s_memtime ticks, cyc = calibrated shader cycles), w = waves per CU ==\n");
    row<V_ADD_F32>("", 32, true); row<V_FMA_F32>("", 32, true); row<V_FMA_F64>("", 32, true); row<V_ADD_F64>("", 32, true); row<V_MUL_F64>("", 32, true);
    row<V_PK_FMA_F32>("", 32, true); row<V_PK_ADD_F32>("", 32, true); row<V_MAX3_U32>("", 32, true); row<V_CNDMASK>("", 32, true); row<V_PERM>("", 32, true);
    row<V_PERMLANE32_SWAP>("", 32, true); row<V_PERMLANE16_SWAP>("", 32, true); row<V_MOV_DPP_ROR8>("", 32, true); row<V_MOV_DPP_QUAD>("", 32, true);
    row<V_CVT_F32_F64>("", 32, true); row<V_CVT_F64_F32>("", 32, true);
    printf("\n== LDS: cost per wave-instruction of the CU's ONE LDS pipe (8 per wait), w = active waves per CU ==\n");
    row<DS_W128>("", 8, false); row<DS_W64>("", 8, false); row<DS_W32>("", 8, false); row<DS_WADDTID>("", 8, false);
    row<DS_R128>("", 8, false); row<DS_R64>("", 8, false); row<DS_R32>("", 8, false); row<DS_BPERM>("(dependent chain of 8)", 8, false);
    printf("\n== LDS stores from ONE half of the CU only (waves on SIMD 0 and 1 active, 16 resident) vs alternating halves (SIMD 0 and 2) ==\n");
    { static const int w16[] = {16}; row<DS_W128>("SIMD {0,1}", 8, false, 0x3, w16, 1); row<DS_W128>("SIMD {0,2}", 8, false, 0x5, w16, 1); row<DS_W128>("SIMD {0}", 8, false, 0x1, w16, 1);
      row<DS_W64>("SIMD {0,1}", 8, false, 0x3, w16, 1); row<DS_W64>("SIMD {0,2}", 8, false, 0x5, w16, 1);
      row<DS_WADDTID>("SIMD {0,1}", 8, false, 0x3, w16, 1); row<DS_WADDTID>("SIMD {0,2}", 8, false, 0x5, w16, 1); }
    printf("\n== exchanges and whole frames: cost PER ITERATION per CU (ticks / cycles / ns of CU time per exchange or per frame) ==\n");
    row<XCH_F64>("per exchange", 1, false); row<XCH_F32>("per exchange", 1, false);
    row<FRAME_LDS>("per frame", 1, false); row<FRAME_MIX>("per frame", 1, false);
    return 0;
}
