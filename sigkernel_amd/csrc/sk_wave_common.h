// sk_wave_common.h -- device helpers shared by the tiled wavefront kernels (sk_wave.hip, sk_wave_adj.hip).
#pragma once
#include "sk_internal.h"

namespace sk {
namespace {

#ifndef SK_WAVE_DEFINED
#define SK_WAVE_DEFINED
constexpr int WAVE = 64;
#endif
constexpr int LINE_UNITS = 8;  // 16-byte units per 128-byte line

typedef __attribute__((address_space(3))) void lds_void;


__device__ __forceinline__ double dpp_shr1(double v, double fill) {
    // lane l receives lane l-1's value; lane 0 keeps `fill`
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// lane l receives lane l-1's value; lane 0 receives 0.0 (bound_ctrl: no `old` operand to set up, two instructions in all)
__device__ __forceinline__ double dpp_shr1_zero(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lane l receives lane l-1's value; lane 0 receives 1.0 (the low word by bound_ctrl, the high word from `old`: three instructions)
__device__ __forceinline__ double dpp_shr1_one(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0x3ff00000, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double dpp_shl1(double v, double fill) {
    // lane l receives lane l+1's value; lane 63 keeps `fill`
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int floor_div(int a, int b) {  // b > 0
    int q = a / b;
    return (a % b < 0) ? q - 1 : q;
}

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Unit;  // one 16-byte unit of increments
template <> struct Unit<double> { static constexpr int CW = 2; typedef d2_t vec; };
template <> struct Unit<float> { static constexpr int CW = 4; typedef f4_t vec; };

template <typename V> __device__ __forceinline__ double vec_get(const V &v, int i) { return (double)v[i]; }

// All LDS traffic of the sweep goes through inline asm.  hipcc cannot tell that a ds_read does not alias
// an LDS-DMA still in flight and would drain the whole prefetch ring with s_waitcnt vmcnt(0) before every
// read; here the DMA queue is counted by hand (vmcnt(N) = fetches still allowed in flight) and the asm
// block itself waits for its own reads (lgkmcnt(0)) before any output is consumed.
template <int VM, typename V>
__device__ __forceinline__ void lds_read_rows(V (&g)[1], unsigned a) {
    asm volatile("s_waitcnt vmcnt(%2)\n\t"
                 "ds_read_b128 %0, %1\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]) : "v"(a), "n"(VM) : "memory");
}
template <int VM, typename V>
__device__ __forceinline__ void lds_read_rows(V (&g)[2], unsigned a) {
    asm volatile("s_waitcnt vmcnt(%3)\n\t"
                 "ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %2 offset:1024\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]) : "v"(a), "n"(VM) : "memory");
}
template <int VM, typename V>
__device__ __forceinline__ void lds_read_rows(V (&g)[4], unsigned a) {
    asm volatile("s_waitcnt vmcnt(%5)\n\t"
                 "ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:1024\n\t"
                 "ds_read_b128 %2, %4 offset:2048\n\t"
                 "ds_read_b128 %3, %4 offset:3072\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]) : "v"(a), "n"(VM) : "memory");
}
// two independent row sets (the increments of this macro-step and a finished W line), one wait
template <typename V>
__device__ __forceinline__ void lds_read_rows_pair(V (&g)[1], unsigned a, V (&w)[1], unsigned b) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(w[0]) : "v"(a), "v"(b) : "memory");
}
template <typename V>
__device__ __forceinline__ void lds_read_rows_pair(V (&g)[2], unsigned a, V (&w)[2], unsigned b) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                 "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(w[0]), "=&v"(w[1]) : "v"(a), "v"(b) : "memory");
}
template <typename V>
__device__ __forceinline__ void lds_read_rows_pair(V (&g)[4], unsigned a, V (&w)[4], unsigned b) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                 "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:1024\n\t"
                 "ds_read_b128 %6, %9 offset:2048\n\tds_read_b128 %7, %9 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                 : "v"(a), "v"(b) : "memory");
}
__device__ __forceinline__ double lds_read_f64(unsigned addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
// four independent 8-byte reads, one wait (a top lane reading its boundary row pays one LDS round trip, not four)
__device__ __forceinline__ void lds_read_f64x4(double &v0, double &v1, double &v2, double &v3, unsigned a0, unsigned a1,
                                               unsigned a2, unsigned a3) {
    asm volatile("ds_read_b64 %0, %4\n\t"
                 "ds_read_b64 %1, %5\n\t"
                 "ds_read_b64 %2, %6\n\t"
                 "ds_read_b64 %3, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                 : "memory");
}
// N consecutive doubles starting at `addr` (N a multiple of 4)
template <int N>
__device__ __forceinline__ void lds_read_row(double (&v)[N], unsigned addr) {
    static_assert(N % 4 == 0, "boundary rows are read four values at a time");
#pragma unroll
    for (int i = 0; i < N; i += 4)
        lds_read_f64x4(v[i], v[i + 1], v[i + 2], v[i + 3], addr + i * 8u, addr + (i + 1) * 8u, addr + (i + 2) * 8u,
                       addr + (i + 3) * 8u);
}
// N consecutive doubles (16-byte aligned) with N/2 ds_read_b128 and ONE wait
template <int N>
__device__ __forceinline__ void lds_read_row1(double (&v)[N], unsigned addr);
template <>
__device__ __forceinline__ void lds_read_row1<2>(double (&v)[2], unsigned p) {
    d2_t t;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(p) : "memory");
    v[0] = t[0]; v[1] = t[1];
}
template <>
__device__ __forceinline__ void lds_read_row1<4>(double (&v)[4], unsigned p) {
    d2_t t[2];
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]) : "v"(p) : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) { v[2 * i] = t[i][0]; v[2 * i + 1] = t[i][1]; }
}
template <>
__device__ __forceinline__ void lds_read_row1<8>(double (&v)[8], unsigned p) {
    d2_t t[4];
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]) : "v"(p) : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = t[i][0]; v[2 * i + 1] = t[i][1]; }
}
template <>
__device__ __forceinline__ void lds_read_row1<16>(double (&v)[16], unsigned p) {
    d2_t t[8];
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
                 "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                 "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                 : "v"(p) : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[2 * i] = t[i][0]; v[2 * i + 1] = t[i][1]; }
}
template <>
__device__ __forceinline__ void lds_read_row1<32>(double (&v)[32], unsigned p) {
    double lo[16], hi[16];
    lds_read_row1<16>(lo, p);
    lds_read_row1<16>(hi, p + 128u);
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = lo[i]; v[16 + i] = hi[i]; }
}

// N consecutive doubles (N even, 16-byte aligned), BLOCKING: every piece is one asm with its own wait inside.  For the kernel variants
// beyond 256 VGPRs (one wave per SIMD, registers spilling to AGPRs), where a read left in flight across other code is unsafe: the
// allocator is free to copy the destination of a separate-asm read before the separate-asm wait (seen as last-bit, run-to-run
// differences; the layout-order lint of tools/check_async_hazards.py does not see it).
template <int N>
__device__ __forceinline__ void lds_read_block(double *v, unsigned a) {
    static_assert(N % 2 == 0 && N >= 0, "");
    if constexpr (N >= 16) {
        double t[16];
        lds_read_row1<16>(t, a);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = t[i];
        lds_read_block<N - 16>(v + 16, a + 128u);
    } else if constexpr (N >= 8) {
        double t[8];
        lds_read_row1<8>(t, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = t[i];
        lds_read_block<N - 8>(v + 8, a + 64u);
    } else if constexpr (N >= 4) {
        double t[4];
        lds_read_row1<4>(t, a);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = t[i];
        lds_read_block<N - 4>(v + 4, a + 32u);
    } else if constexpr (N >= 2) {
        double t[2];
        lds_read_row1<2>(t, a);
        v[0] = t[0]; v[1] = t[1];
    }
}
// N consecutive doubles at an 8-byte aligned address, BLOCKING (one asm, one wait)
template <int N>
__device__ __forceinline__ void lds_read_f64_block(double (&t)[N], unsigned a);
template <>
__device__ __forceinline__ void lds_read_f64_block<2>(double (&t)[2], unsigned a) {
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t[0]), "=&v"(t[1]) : "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_f64_block<4>(double (&t)[4], unsigned a) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]) : "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_f64_block<8>(double (&t)[8], unsigned a) {
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                 "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                 : "v"(a) : "memory");
}

// two rows of N consecutive doubles each (the Kr / Kf band boundaries of the adjoint), all reads issued before ONE wait
template <int N>
__device__ __forceinline__ void lds_read_2rows(double (&a)[N], double (&b)[N], unsigned addr_a, unsigned addr_b);
template <>
__device__ __forceinline__ void lds_read_2rows<2>(double (&a)[2], double (&b)[2], unsigned pa, unsigned pb) {
    d2_t t[2];
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]) : "v"(pa), "v"(pb) : "memory");
    a[0] = t[0][0]; a[1] = t[0][1]; b[0] = t[1][0]; b[1] = t[1][1];
}
template <>
__device__ __forceinline__ void lds_read_2rows<4>(double (&a)[4], double (&b)[4], unsigned pa, unsigned pb) {
    d2_t t[4];
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]) : "v"(pa), "v"(pb) : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) { a[2 * i] = t[i][0]; a[2 * i + 1] = t[i][1]; b[2 * i] = t[2 + i][0]; b[2 * i + 1] = t[2 + i][1]; }
}
template <>
__device__ __forceinline__ void lds_read_2rows<8>(double (&a)[8], double (&b)[8], unsigned pa, unsigned pb) {
    d2_t t[8];
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
                 "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:16\n\t"
                 "ds_read_b128 %6, %9 offset:32\n\tds_read_b128 %7, %9 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                 : "v"(pa), "v"(pb) : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[2 * i] = t[i][0]; a[2 * i + 1] = t[i][1]; b[2 * i] = t[4 + i][0]; b[2 * i + 1] = t[4 + i][1]; }
}

// N consecutive doubles starting at `addr` (8-byte aligned), issued WITHOUT a wait: the destinations are temporaries nobody
// reads until lds_take hands them over after a later s_waitcnt lgkmcnt(0) (same discipline as load_async / async_wait)
template <int N>
__device__ __forceinline__ void lds_read_f64_run(double (&t)[N], unsigned a);
template <>
__device__ __forceinline__ void lds_read_f64_run<2>(double (&t)[2], unsigned a) {
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=&v"(t[0]), "=&v"(t[1]) : "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_f64_run<4>(double (&t)[4], unsigned a) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]) : "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_f64_run<8>(double (&t)[8], unsigned a) {
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                 "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                 : "v"(a) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_take(double (&o)[N], double (&t)[N]);
template <>
__device__ __forceinline__ void lds_take<2>(double (&o)[2], double (&t)[2]) {
    asm volatile("" : "=v"(o[0]), "=v"(o[1]) : "0"(t[0]), "1"(t[1]));
}
template <>
__device__ __forceinline__ void lds_take<4>(double (&o)[4], double (&t)[4]) {
    asm volatile("" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]) : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]));
}
template <>
__device__ __forceinline__ void lds_take<8>(double (&o)[8], double (&t)[8]) {
    asm volatile("" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7])
                 : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]), "4"(t[4]), "5"(t[5]), "6"(t[6]), "7"(t[7]));
}

__device__ __forceinline__ void lds_write_f64(unsigned addr, double v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <typename V>
__device__ __forceinline__ void lds_write_b128(unsigned addr, V v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_offset(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}

template <int DY> struct Tile {
    static constexpr int RC = DY == 0 ? 4 : DY == 1 ? 2 : 1;  // coarse rows per lane
    static constexpr int R = RC << DY;                         // fine rows per lane
};

// (tuning knobs: struct Knobs / knobs() in sk_internal.h)

// ---- hand-managed asynchronous global loads (sk_wave_adj.hip, sk_wave_adj_fused.hip) ----------------------------------
// Asynchronous 8-byte global loads into registers.  The compiler must never touch a destination register between the
// load and the wait (it does not know the value is still in flight -- its own waitcnt insertion would drain every
// LDS-DMA at the first use, which is why these are asm).  So the destinations are short-lived temporaries that nothing
// reads: async_begin() defines them (no instruction), load_async() may or may not overwrite them (conditional code:
// the merge is with an equally unread value, so no copy is needed), and async_wait() is the single instruction that
// turns them into ordinary values -- its outputs are tied to the temporaries' registers.  tests/test_abi.py scans the
// generated ISA for any instruction that touches a pending destination (tools/check_async_hazards.py).
__device__ __forceinline__ void async_begin(double &t) { asm volatile("" : "=v"(t)); }
__device__ __forceinline__ void load_async(double &dst, const double *p) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int BYTE_OFF>   // immediate offset (13-bit signed): one address register pair serves a run of loads
__device__ __forceinline__ void load_async_at(double &dst, const double *p) {
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(BYTE_OFF) : "memory");
}
// dst[i] <- p[-i], i = 0 .. N-1
template <int N, int M>
__device__ __forceinline__ void load_run(double (&dst)[M], const double *p) {
    static_assert(N <= M && N <= 8, "");
    if constexpr (N > 0) load_async_at<0>(dst[0], p);
    if constexpr (N > 1) load_async_at<-8>(dst[1], p);
    if constexpr (N > 2) load_async_at<-16>(dst[2], p);
    if constexpr (N > 3) load_async_at<-24>(dst[3], p);
    if constexpr (N > 4) load_async_at<-32>(dst[4], p);
    if constexpr (N > 5) load_async_at<-40>(dst[5], p);
    if constexpr (N > 6) load_async_at<-48>(dst[6], p);
    if constexpr (N > 7) load_async_at<-56>(dst[7], p);
}

template <int VM>
__device__ __forceinline__ void async_wait(double (&o)[4], double (&t)[4]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3])
                 : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait(double (&o)[1], double (&t)[1]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "=v"(o[0]) : "0"(t[0]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait(double (&o)[2], double (&t)[2]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "=v"(o[0]), "=v"(o[1]) : "0"(t[0]), "1"(t[1]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait(double (&o)[3], double (&t)[3]) {
    asm volatile("s_waitcnt vmcnt(%6)" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]) : "0"(t[0]), "1"(t[1]), "2"(t[2]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait(double (&o)[5], double (&t)[5]) {
    asm volatile("s_waitcnt vmcnt(%10)" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4])
                 : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]), "4"(t[4]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait(double (&o)[8], double (&t)[8]) {
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7])
                 : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]), "4"(t[4]), "5"(t[5]), "6"(t[6]), "7"(t[7]), "n"(VM) : "memory");
}

// 1/b for b = 1 - g^2/12 (close to 1): hardware estimate + two Newton steps, instead of the ~20-instruction IEEE
// division sequence.  The result only has to be a consistent multiplier: a/b and 1/b are formed from the same value.
__device__ __forceinline__ double fast_rcp(double b) {
    double x = __builtin_amdgcn_rcp(b);
    x = fma(x, fma(-b, x, 1.0), x);
    x = fma(x, fma(-b, x, 1.0), x);
    return x;
}


// ---- workgroups of independent waves -----------------------------------------------------------------------------------
// Every wavefront kernel here is a set of persistent, mutually independent waves, each with a private LDS slice and no
// barrier.  They are nevertheless launched as workgroups of up to four waves: the dispatcher deals a workgroup's waves
// round-robin over the CU's four SIMDs, whereas single-wave workgroups land wherever there is room and leave the SIMDs
// unevenly loaded (measured on the fused kernel: 10 single-wave workgroups per CU 6.3 ms, 3 x 4 waves 5.3 ms).
struct WaveGroup {
    int wpb;            // waves per workgroup (1..4)
    int lds_per_wave;   // bytes of dynamic LDS of one wave
    int64_t n_waves;    // waves in the launch; the last workgroup may be partial
};

// host: waves per workgroup, and a resident-waves-per-CU figure rounded to whole workgroups
inline WaveGroup wave_group(size_t lds_per_wave, int64_t n_waves, int knob, int dflt = 4) {
    int wpb = knob > 0 ? knob : dflt;
    if (wpb < 1) wpb = 1;
    if (wpb > 4) wpb = 4;
    while (wpb > 1 && (size_t)wpb * lds_per_wave > 160 * 1024) --wpb;
    while (wpb > 1 && n_waves < wpb) --wpb;
    return WaveGroup{wpb, (int)lds_per_wave, n_waves};
}
inline int wave_group_blocks(const WaveGroup &wg) { return (int)((wg.n_waves + wg.wpb - 1) / wg.wpb); }
inline size_t wave_group_lds(const WaveGroup &wg) { return (size_t)wg.wpb * (size_t)wg.lds_per_wave; }

// device: this wave's index in the launch (-1: beyond the end) and its LDS slice
__device__ __forceinline__ int64_t wave_slot(const WaveGroup &wg, char *lds_block, char *&lds) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    lds = lds_block + (size_t)wv * wg.lds_per_wave;
    const int64_t id = (int64_t)blockIdx.x * wg.wpb + wv;
    return id < wg.n_waves ? id : -1;
}

// ---- unequal shares for the waves that share a SIMD ---------------------------------------------------------------------
// The SIMD arbiter favours the OLDEST resident wave: measured on the fused forward (3 four-wave workgroups per CU, equal
// shares, profiles/r02_wave_placement.txt) the three waves of every SIMD finished after 2.8, 4.0 and 5.0 ms -- the SIMD ran
// with two waves for a fifth of the launch and with one for another fifth.  The dispatcher fills the CUs in workgroup order
// (workgroup i, i + #CU, i + 2 #CU share a CU; verified from HW_ID on all 1024 SIMDs), so a wave's age rank is
// blockIdx / #CU, and the launchers give rank r the fraction w[r] of the pairs, chosen so that all ranks finish together.
// A wrong guess about placement costs speed, never correctness: the split is a partition of the pairs whatever the ranks.
// (device_cu_count(): 256 on MI355X; the launchers size their persistent launches by it, and on a launch that does not consist
// of whole ranks the shares below fall back to equal ones by themselves)
// a family's SK_*_RANK_W (else the global SK_RANK_W) replaces the default shares when it names exactly nr ranks
inline void rank_override(const RankW &family, int nr, double (&w)[4]) {
    const RankW &o = family.n > 0 ? family : knobs().rank_w;
    double tot = 0;
    for (int r = 0; r < o.n && r < 4; ++r) tot += o.w[r];
    if (o.n == nr && tot > 0)
        for (int r = 0; r < nr; ++r) w[r] = o.w[r] / tot;
}

struct RankSplit {
    int nranks;              // 1: every wave gets cnt[0] pairs per lane group (the plain equal split)
    int waves_per_rank;      // #CU * waves per workgroup
    int cnt[4];              // pairs per lane group of a wave of rank r
    int64_t base[5];         // first pair of rank r; base[nranks] >= P
};

// host: split P pairs over `waves` waves of G lane groups each.  `resident` is the number of waves the launch keeps on the
// chip at once (#CU * waves per CU); ranks are only used when the launch fills it (waves == resident) with whole workgroups.
// `table` (optional): the kernel family's own default shares, rows indexed by the number of ranks.
inline RankSplit rank_split(int64_t P, int G, int64_t waves, int64_t resident, int wpb, int n_cu, const RankW &family,
                            const double (*table)[4] = nullptr) {
    RankSplit rs{};
    const int64_t per = (P + waves * G - 1) / (waves * G);
    rs.nranks = 1; rs.waves_per_rank = 0x7fffffff; rs.cnt[0] = (int)per; rs.base[0] = 0; rs.base[1] = per * waves * G;
    const int64_t wpr = (int64_t)n_cu * wpb;
    const int nr = wpr > 0 ? (int)(waves / wpr) : 0;
    if (waves != resident || nr < 2 || nr > 4 || (int64_t)nr * wpr != waves) return rs;
    // measured finishing times with equal shares, per number of ranks (see above); SK_RANK_W="50,30,20" overrides (per cent)
    static constexpr double dflt[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0.66, 0.34, 0, 0}, {0.53, 0.30, 0.17, 0}, {0.40, 0.27, 0.19, 0.14}};
    double w[4];
    for (int r = 0; r < 4; ++r) w[r] = table ? table[nr][r] : dflt[nr][r];
    rank_override(family, nr, w);
    const int64_t T = (P + wpr * G - 1) / (wpr * G);   // pairs per lane group summed over the ranks of one SIMD slot
    if (T < 4 * nr) return rs;                          // too few pairs for the split to matter
    int64_t used = 0;
    rs.nranks = nr; rs.waves_per_rank = (int)wpr;
    for (int r = 0; r < nr; ++r) {
        int64_t c = r + 1 < nr ? (int64_t)(w[r] * (double)T + 0.5) : T - used;
        if (c < 1) c = 1;
        if (r + 1 == nr && c < 1) c = 1;
        rs.cnt[r] = (int)c;
        rs.base[r] = used * wpr * G;
        used += c;
    }
    rs.base[nr] = used * wpr * G;
    return rs;
}

// device: this wave's share -- pairs per lane group, first pair of its lane group 0, and the end of its rank's range
__host__ __device__ __forceinline__ void rank_share(const RankSplit &rs, int64_t wave_id, int G, int64_t P, int &ppg, int64_t &first, int64_t &end) {
    int r = (int)(wave_id / rs.waves_per_rank);
    if (r >= rs.nranks) r = rs.nranks - 1;
    ppg = rs.cnt[0]; first = rs.base[0]; end = rs.base[1];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (r == i) { ppg = rs.cnt[i]; first = rs.base[i]; end = rs.base[i + 1]; }
    first += (wave_id - (int64_t)r * rs.waves_per_rank) * G * ppg;
    if (end > P) end = P;
}

// The fused adjoints split a Gram differently: a lane group sweeps one CHUNK of the B pairs of one path x_a and leaves a
// partial sum in slot a * nch + c, which the host adds over c.  Chunks need not be equal for that, so the chunks swept by the
// oldest waves are made longer: chunk c of every a belongs to rank c / cpr, and rank r's chunks hold size[r] pairs.
// (struct ChunkSplit: sk_internal.h)

// Pairs per lane-group chunk of a fused adjoint's Gram launch (B > 0): the smallest divisor of B that keeps A B / d lane groups within
// the resident ones -- or, when that leaves more of them idle, the uneven split into as many chunks per a as they take (ChunkSplit::uneven).
inline int64_t pick_chunk(int64_t A, int64_t B, int64_t max_groups) {
    if (B <= 0) return 1;
    int64_t ppg = B;
    for (int64_t d = 1; d <= B; ++d)
        if (B % d == 0 && A * (B / d) <= max_groups) { ppg = d; break; }
    int64_t nch = max_groups / (A > 0 ? A : 1);
    if (nch < 1) nch = 1;
    if (nch > B) nch = B;
    const int64_t ppg_u = (B + nch - 1) / nch;
    return ppg_u < ppg ? ppg_u : ppg;      // (shorter chunks = more lane groups at work; equal when a divisor does as well)
}
inline int64_t chunks_of(int64_t B, int64_t ppg) { return B > 0 ? (B + ppg - 1) / ppg : 1; }

inline ChunkSplit chunk_split(int64_t A, int64_t B, int64_t PPG, int64_t max_groups, int G, int wpb, int n_cu, const RankW &family) {
    ChunkSplit cs{};
    // PPG: the LONGEST chunk; nch = ceil(B / PPG) chunks per a, of B / nch pairs each, rounded down or up
    const int64_t nch = B > 0 ? (B + PPG - 1) / PPG : 1;
    cs.nr = 1; cs.cpr = (int)nch; cs.nch = (int)nch; cs.gpr = (int64_t)1 << 62; cs.size[0] = (int)(B > 0 ? (B + nch - 1) / nch : PPG); cs.off[0] = 0;
    cs.B = (int)B; cs.uneven = B > 0 && B % nch != 0;
    const int64_t gpr = (int64_t)n_cu * wpb * G;
    if (B <= 0 || gpr <= 0 || cs.uneven || A * nch != max_groups || max_groups % gpr) return cs;
    const int nr = (int)(max_groups / gpr);
    if (nr < 2 || nr > 4 || nch % nr || PPG < 4 * nr) return cs;
    static constexpr double dflt[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0.66, 0.34, 0, 0}, {0.53, 0.30, 0.17, 0}, {0.40, 0.27, 0.19, 0.14}};
    double w[4];
    for (int r = 0; r < 4; ++r) w[r] = dflt[nr][r];
    rank_override(family, nr, w);
    const int cpr = (int)(nch / nr);
    const int64_t T = B / cpr;            // pairs of one a per "column" of ranks: sum of the sizes
    int64_t used = 0;
    for (int r = 0; r < nr; ++r) {
        int64_t c = r + 1 < nr ? (int64_t)(w[r] * (double)T + 0.5) : T - used;
        if (c < 1 || used + c > T - (nr - 1 - r)) return cs;   // (degenerate weights: keep the equal chunks)
        cs.size[r] = (int)c;
        cs.off[r] = (int)(used * cpr);
        used += c;
    }
    cs.nr = nr; cs.cpr = cpr; cs.gpr = gpr;
    return cs;
}

// device (per lane): lane group gi -> its first pair, its slot in the partial-sum array, and the pairs it sweeps
__host__ __device__ __forceinline__ void chunk_share(const ChunkSplit &cs, int64_t gi, int64_t A, int64_t B, int64_t P, int64_t &first, int64_t &slot, int &ppg) {
    if (B <= 0) {      // paired batch: ppp consecutive pairs per lane group (one until round 6), every pair with a slot of its own
        const int64_t ppp = cs.ppp > 1 ? cs.ppp : 1;
        first = gi * ppp < P ? gi * ppp : P;
        slot = first;
        ppg = (int)(P - first < ppp ? P - first : ppp);
        return;
    }
    if (cs.uneven) {     // nch does not divide B: chunk c of an a is [c B / nch, (c + 1) B / nch)
        const int64_t a = gi / cs.nch;
        const int64_t cl = gi - a * cs.nch;
        const int64_t lo = cl * cs.B / cs.nch, hi = (cl + 1) * cs.B / cs.nch;
        ppg = (int)(hi - lo);
        slot = gi;
        first = a < A ? a * B + lo : P;
        return;
    }
    int r = (int)(gi / cs.gpr);
    if (r >= cs.nr) r = cs.nr - 1;
    int sz = cs.size[0], of = cs.off[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (r == i) { sz = cs.size[i]; of = cs.off[i]; }
    const int64_t j = gi - (int64_t)r * cs.gpr;
    const int64_t a = j / cs.cpr;
    const int cl = (int)(j - a * cs.cpr);
    ppg = sz;
    slot = a * cs.nch + (int64_t)r * cs.cpr + cl;
    first = a < A ? a * B + of + (int64_t)cl * sz : P;
}
__device__ __forceinline__ int64_t gather64(int64_t v, int src_lane) {   // per-lane source (ds_bpermute: no LDS memory involved)
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(v >> 32));
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int64_t readlane64(int64_t v, int lane) {
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane), hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

}  // namespace
}  // namespace sk
