// Victim kernel of tools/pk_opsel_repro.py: the packed-fp32 FMA of the LN-fold epilogue in isolation.
//   form 1: v_pk_fma_f32 D, acc, ms, C op_sel:[0,1,0]   (high half of SRC1 into the low lane: the form measured wrong)
//   form 2: v_pk_fma_f32 D, ms, acc, C op_sel:[1,0,0]   (high half of SRC0 into the low lane: the form that never failed)
//   forms 3-8: SRC2 high half, v_pk_mul / v_pk_add with SRC1 high half, SRC1 low broadcast (op_sel_hi = 0), SRC1 halves swapped,
//              v_pk_mul with SRC0 high half -- see the kernel
//   form + 10 (MIX): the odd waves of every workgroup issue back-to-back MFMAs instead, so MFMA waves and checking waves share SIMDs by
//              construction, whatever else runs on the chip
// Every result is compared, bit for bit, with two scalar v_fma_f32 of the same operands; mismatching lane-iterations are counted.
// Light on purpose (no MFMA, 8 KB of LDS, < 64 VGPRs): its workgroups co-reside with whatever else runs on the CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_v;
typedef float f32x4_v __attribute__((ext_vector_type(4)));

template <int FORM, bool MIX>
__global__ __launch_bounds__(256) void victim_kernel(int iters, unsigned long long *mism, unsigned long long *first_bad) {
    __shared__ f32x2 table[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) {
        const float mean = 0.001f * (float)((i * 37) % 997) - 0.5f, rstd = 0.5f + 0.002f * (float)((i * 61) % 1009);
        table[i] = f32x2{mean, rstd};
    }
    __syncthreads();
    if (MIX && ((tid >> 6) & 1)) {   // MIX: the odd waves of the SAME workgroup issue MFMAs, the even ones run the spelling under test
        bf16x8_v fa, fb;
        for (int i = 0; i < 8; ++i) {
            fa[i] = (__bf16)(0.01f * (tid + i));
            fb[i] = (__bf16)(0.02f * (tid - i));
        }
        f32x4_v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j], 0, 0, 0);
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) atomicAdd(mism, 1ull);
        return;
    }
    f32x2 a = {1.0f + 0.001f * tid, -0.5f + 0.002f * tid}, c = {0.25f, -0.75f};
    unsigned long long bad = 0, bad_lo = 0;
    uint32_t idx = tid * 7u + blockIdx.x * 13u;
    for (int it = 0; it < iters; ++it) {
        const f32x2 ms = table[idx & 1023u];            // ds_read_b64: (mean, rstd) as the epilogue reads it
        idx = idx * 1664525u + 1013904223u;
        // the register pair as a 64-bit scalar operand (element extraction from a 2-vector asm output read the low register twice)
        unsigned long long r;
        const unsigned long long a64 = __builtin_bit_cast(unsigned long long, a), ms64 = __builtin_bit_cast(unsigned long long, ms),
                                 c64 = __builtin_bit_cast(unsigned long long, c);
        float e0, e1;
        if (FORM == 1) {          // SRC1 high half -> low lane
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a64), "v"(ms64), "v"(c64));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(ms[1]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[1]), "v"(ms[1]), "v"(c[1]));
        } else if (FORM == 2) {   // SRC0 high half -> low lane (same values)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(ms64), "v"(a64), "v"(c64));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(ms[1]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[1]), "v"(ms[1]), "v"(c[1]));
        } else if (FORM == 3) {   // SRC2 high half -> low lane
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(a64), "v"(c64), "v"(ms64));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(c[0]), "v"(ms[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[1]), "v"(c[1]), "v"(ms[1]));
        } else if (FORM == 4) {   // v_pk_mul_f32, SRC1 high half -> low lane
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a64), "v"(ms64));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(ms[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(ms[1]));
        } else if (FORM == 5) {   // v_pk_add_f32, SRC1 high half -> low lane
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a64), "v"(ms64));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(ms[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(ms[1]));
        } else if (FORM == 6) {   // SRC1 LOW half broadcast by op_sel_hi = 0 (what the compiler emits for a scalar in a low half)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a64), "v"(ms64), "v"(c64));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(ms[0]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[1]), "v"(ms[0]), "v"(c[1]));
        } else if (FORM == 7) {   // SRC1 halves swapped: high -> low lane, low -> high lane
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a64), "v"(ms64), "v"(c64));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(ms[1]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[1]), "v"(ms[0]), "v"(c[1]));
        } else {                  // 8: v_pk_mul_f32, SRC0 high half -> low lane
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(ms64), "v"(a64));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(ms[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(ms[1]));
        }
        const bool b = (uint32_t)r != __builtin_bit_cast(uint32_t, e0) || (uint32_t)(r >> 32) != __builtin_bit_cast(uint32_t, e1);
        const bool blo = (uint32_t)r != __builtin_bit_cast(uint32_t, e0);
        bad_lo += blo ? 1ull : 0ull;
        bad += b ? 1ull : 0ull;
        // keep the operands moving and finite
        a[0] = a[0] * 0.999f + 0.013f;
        a[1] = a[1] * 0.998f - 0.007f;
        c[0] = (FORM == 4 || FORM == 5 || FORM == 8 ? c[0] * 0.9f : e0 * 0.5f) + 0.1f;
        c[1] = (FORM == 4 || FORM == 5 || FORM == 8 ? c[1] * 0.9f : e1 * 0.5f) - 0.1f;
    }
    if (bad) {
        atomicAdd(mism, bad);
        atomicAdd(first_bad, bad_lo);      // second counter: mismatches whose LOW lane is wrong
    }
}

extern "C" int pk_victim_launch(int form, int blocks, int iters, unsigned long long *mism, unsigned long long *first_bad, void *stream) {
// form 1..8 = the spelling alone in the kernel; form + 10 = the same with MFMA-issuing odd waves in the same workgroups
#define PK_CASE(F) case F: hipLaunchKernelGGL((victim_kernel<F, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, mism, first_bad); break; \
                   case F + 10: hipLaunchKernelGGL((victim_kernel<F, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, mism, first_bad); break;
    switch (form) {
        PK_CASE(1) PK_CASE(2) PK_CASE(3) PK_CASE(4) PK_CASE(5) PK_CASE(6) PK_CASE(7) PK_CASE(8)
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ---- synthetic co-tenants (tools/pk_opsel_repro.py --synthetic): one instruction class each, 256 threads, 32 KB of LDS, < 128 VGPRs, so that
// their workgroups share CUs (and SIMDs) with the victim's.  mode: 1 MFMA only | 2 LDS-DMA (global_load_lds) | 3 ds_read_b128 |
// 4 packed-fp32 VALU | 5 global loads to VGPRs + stores | 6 MFMA fed by ds_read_b128 | 7 s_barrier | 8 ds_write_b64 + ds_read_b128
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__global__ __launch_bounds__(256) void cotenant_kernel(int mode, int iters, const float *__restrict__ src, float *__restrict__ dst, size_t n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * 4;
    f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8_t fa, fb;
    for (int i = 0; i < 8; ++i) {
        fa[i] = (__bf16)(0.01f * (tid + i));
        fb[i] = (__bf16)(0.02f * (tid - i));
    }
    float4 *l4 = (float4 *)smem;
    for (int i = tid; i < 2048; i += 256) l4[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x4_t v = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
        if (mode == 1 || mode == 6) {
            if (mode == 6) {
                const float4 t = l4[(tid + it * 17) & 2047];
                fa = __builtin_bit_cast(bf16x8_t, t);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j], 0, 0, 0);
        } else if (mode == 2) {
            const size_t off = (base + (size_t)it * 262144) % (n - 1024);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (off & ~(size_t)3)), (lds_void_t *)(smem + (it & 7) * 4096), 16, 0, 0);
            if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (mode == 3) {
            const float4 t = l4[(tid * 3 + it) & 2047];
            v[0] += t.x;
            v[1] += t.y;
        } else if (mode == 4) {
            v = v * 1.0001f + 0.5f;
        } else if (mode == 5) {
            const size_t off = (base + (size_t)it * 262144) % (n - 1024);
            const float4 t = *(const float4 *)(src + (off & ~(size_t)3));
            *(float4 *)(dst + (off & ~(size_t)3)) = make_float4(t.x + 1.f, t.y, t.z, t.w);
        } else if (mode == 7) {
            __syncthreads();
            v[0] += 1.f;
        } else if (mode == 8) {
            *(float2 *)(smem + ((tid * 8 + it * 64) & 32767)) = make_float2(v[0], v[1]);
            const float4 t = l4[(tid * 5 + it) & 2047];
            v[2] += t.z;
        }
    }
    if (mode == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v[0] + v[1] + v[2] + v[3];
    if (s == 12345.678f) dst[tid] = s;
}

extern "C" int pk_cotenant_launch(int mode, int blocks, int iters, const float *src, float *dst, size_t n, void *stream) {
    hipLaunchKernelGGL(cotenant_kernel, dim3(blocks), dim3(256), 32768, (hipStream_t)stream, mode, iters, src, dst, n);
    return (int)hipGetLastError();
}

// ---- exactness of OTHER instruction classes next to MFMA-issuing waves (tools/pk_opsel_repro.py --classes) -------------------------------
// The even waves of every workgroup run `iters` iterations of one instruction class on fixed inputs and leave a 64-bit checksum per lane; the
// odd waves either exit at once (MIX = false) or issue back-to-back MFMAs (MIX = true).  Same inputs, same instruction stream: any checksum
// that differs between the two launches is the hardware.  OP: 1 v_pk_fma_f32 (no op_sel) | 2 v_pk_mul_f32 op_sel_hi:[1,0] + v_pk_add_f32
// op_sel_hi:[0,1] | 3 v_cvt_pk_bf16_f32 | 4 v_exp_f32 + v_rcp_f32 | 5 DPP: v_mov_b32_dpp quad_perm + v_xor_b32_dpp row_newbcast | 6 v_bcnt_u32_b32
// chain | 7 ds_write_b64 + ds_read_b128 round trip | 8 integer VALU (v_add_u32, v_lshl_or_b32, v_min_u32 / v_max_u32, v_cndmask) |
// 9 scalar v_fma_f32 / v_mul / v_add | 10 POSITIVE CONTROL: v_pk_fma_f32 op_sel:[0,1,0]
template <int OP, bool MIX>
__global__ __launch_bounds__(256) void class_kernel(int iters, unsigned long long *__restrict__ sums) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 8];
    const int tid = threadIdx.x;
    if ((tid >> 6) & 1) {
        if (!MIX) return;
        bf16x8_v fa, fb;
        for (int i = 0; i < 8; ++i) {
            fa[i] = (__bf16)(0.01f * (tid + i));
            fb[i] = (__bf16)(0.02f * (tid - i));
        }
        f32x4_v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j], 0, 0, 0);
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sums[0] = 1;
        return;
    }
    unsigned long long sum = 0;
    for (int i = 0; i < 8; ++i) lds[tid * 8 + i] = 0.5f * i;     // a lane only ever touches its own eight floats
    f32x2 a = {1.0f + 0.001f * tid, -0.5f + 0.002f * tid}, b = {0.75f + 0.0005f * tid, 1.25f - 0.0007f * tid}, c = {0.25f, -0.75f};
    uint32_t u = 0x9E3779B9u * (tid + 1) + blockIdx.x, w = 0x85EBCA6Bu ^ (uint32_t)tid;
    for (int it = 0; it < iters; ++it) {
        if (OP == 1) {
            unsigned long long r;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(__builtin_bit_cast(unsigned long long, a)), "v"(__builtin_bit_cast(unsigned long long, b)),
                         "v"(__builtin_bit_cast(unsigned long long, c)));
            sum += r;
            c = __builtin_bit_cast(f32x2, r) * 0.5f + 0.1f;
        } else if (OP == 2) {
            unsigned long long r, q;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(__builtin_bit_cast(unsigned long long, a)), "v"(__builtin_bit_cast(unsigned long long, b)));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(q) : "v"(r), "v"(__builtin_bit_cast(unsigned long long, c)));
            sum += q;
            c = __builtin_bit_cast(f32x2, q) * 0.25f;
        } else if (OP == 3) {
            uint32_t r;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[0]), "v"(c[1]));
            sum += r;
            c[1] = c[1] * 0.999f + 0.01f;
        } else if (OP == 4) {
            float e, r;
            asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(c[0]));
            const float d = 1.0f + e;
            asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(d));
            sum += __builtin_bit_cast(uint32_t, r);
            c[0] = r * 2.0f - 1.0f + 0.001f * (it & 15);
        } else if (OP == 5) {
            uint32_t r, q;
            asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(u));
            asm volatile("s_nop 1\n\tv_xor_b32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(q) : "v"(r), "v"(w));
            sum += q;
            u = u * 1664525u + 1013904223u;
            w ^= q >> 3;
        } else if (OP == 6) {
            uint32_t r;
            asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(u), "v"(w & 255u));
            sum += r;
            u = u * 1664525u + 1013904223u;
            w += r;
        } else if (OP == 7) {
            *(float2 *)(lds + tid * 8 + (it & 3) * 2) = make_float2(a[0], c[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const float4 t = *(const float4 *)(lds + tid * 8 + ((it & 1) * 4));
            sum += __builtin_bit_cast(uint32_t, t.x) + __builtin_bit_cast(uint32_t, t.y);
            a[0] = a[0] * 0.999f + 0.013f;
            c[1] = c[1] * 0.998f - 0.007f;
        } else if (OP == 8) {
            const uint32_t k = (u << 9) | (w & 511u);
            const uint32_t lo = k < w ? k : w, hi = k < w ? w : k;
            sum += lo + (hi >> 1);
            u = u * 1664525u + 1013904223u;
            w = w * 22695477u + 1u + (lo & 1u);
        } else if (OP == 9) {
            float r;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a[0]), "v"(b[1]), "v"(c[0]));
            sum += __builtin_bit_cast(uint32_t, r);
            c[0] = r * 0.5f + 0.1f;
        } else {
            unsigned long long r;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(__builtin_bit_cast(unsigned long long, a)), "v"(__builtin_bit_cast(unsigned long long, b)),
                         "v"(__builtin_bit_cast(unsigned long long, c)));
            sum += r;
            c = __builtin_bit_cast(f32x2, r) * 0.5f + 0.1f;
        }
        a[0] = a[0] * 0.9999f + 0.0001f;
    }
    sums[(size_t)blockIdx.x * 256 + tid] = sum;
}

extern "C" int pk_class_launch(int op, int mix, int blocks, int iters, unsigned long long *sums, void *stream) {
#define CL_CASE(O) case O: if (mix) hipLaunchKernelGGL((class_kernel<O, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sums); \
                           else hipLaunchKernelGGL((class_kernel<O, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sums); break;
    switch (op) {
        CL_CASE(1) CL_CASE(2) CL_CASE(3) CL_CASE(4) CL_CASE(5) CL_CASE(6) CL_CASE(7) CL_CASE(8) CL_CASE(9) CL_CASE(10)
        default: return -1;
    }
    return (int)hipGetLastError();
}
