// Hardware probe for gfx950: confirms the MFMA operand/accumulator lane maps, the
// transposing LDS read and the direct global->LDS load that the kernels in
// vision_transformers_torch_xla_amd/csrc rely on.  Build: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static __device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
static __device__ __forceinline__ float bf2f(unsigned short u) {
    return __uint_as_float(((unsigned)u) << 16);
}

// ---- test 1: mfma_f32_16x16x32_bf16 maps.  A[16][32], B[32][16] row-major in global.
__global__ void k_mfma(const unsigned short* A, const unsigned short* B, float* D) {
    int l = threadIdx.x;
    int r = l & 15, g = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[r * 32 + 8 * g + j];          // A[row r][k=8g+j]
        b[j] = (short)B[(8 * g + j) * 16 + r];        // B[k=8g+j][col r]
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(g * 4 + j) * 16 + r] = c[j];   // row=(l>>4)*4+j, col=l&15
}

// ---- test 2: ds_read_b64_tr_b16.  LDS holds a [R][C] u16 image with value = row*C+col.
__global__ void k_tr(unsigned short* out /*[64][4]*/, int rowstride_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];
    int l = threadIdx.x;
    for (int i = l; i < 64 * 64; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    int grp = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    // group grp reads block rows 4*grp .. 4*grp+3, cols 16..31
    const unsigned short* ptr = &img[(4 * grp + q) * rowstride_elems + 16 + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

// ---- test 3: global_load_lds 16 B per lane, lane-linear destination.
__global__ void k_glds(const unsigned* src /*[64*4 + pad]*/, unsigned* out /*[256]*/) {
    __shared__ __attribute__((aligned(16))) unsigned buf[512];
    int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) buf[i] = 0xdeadbeefu;
    __syncthreads();
    // every lane loads 16 B from a permuted source: lane l reads source chunk (l ^ 5)
    const unsigned* g = src + 4 * (l ^ 5);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)&buf[64], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[i] = buf[64 + i];
}

// ---- test 4: f32 -> bf16 conversion is round-to-nearest-even
__global__ void k_cvt(const float* in, unsigned short* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf(in[i]);
}

static unsigned short host_f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float host_bf2f(unsigned short h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    int fails = 0;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s arch=%s CUs=%d lds/block=%zu\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.sharedMemPerBlock);
    {   // test 1
        std::vector<unsigned short> A(16 * 32), B(32 * 16);
        std::vector<float> Af(16 * 32), Bf(32 * 16), D(256), ref(256, 0.f);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { float v = (float)((i * 3 + k * 5) % 7 - 3); Af[i * 32 + k] = v; A[i * 32 + k] = host_f2bf(v); }
        for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) { float v = (float)((k * 2 + j * 7 + k * j) % 9 - 4); Bf[k * 16 + j] = v; B[k * 16 + j] = host_f2bf(v); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += Af[i * 32 + k] * Bf[k * 16 + j]; ref[i * 16 + j] = s; }
        unsigned short *dA, *dB; float* dD;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, 256 * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 256; ++i) if (D[i] != ref[i]) ++bad;
        printf("[mfma16x16x32 maps] mismatches=%d %s\n", bad, bad ? "FAIL" : "OK");
        fails += bad != 0;
    }
    for (int rs : {64, 32}) {   // test 2
        unsigned short* dO; CK(hipMalloc(&dO, 64 * 4 * 2));
        k_tr<<<1, 64>>>(dO, rs); CK(hipDeviceSynchronize());
        std::vector<unsigned short> O(256); CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            int grp = l >> 4, i = l & 15;
            for (int j = 0; j < 4; ++j) {
                int expect = (4 * grp + j) * rs + 16 + i;   // lane i gets column i, row j of the block
                if (O[l * 4 + j] != expect) { if (bad < 8) printf("  tr lane %d elem %d got %d expect %d\n", l, j, O[l * 4 + j], expect); ++bad; }
            }
        }
        printf("[ds_read_b64_tr_b16 rowstride=%d] mismatches=%d %s\n", rs, bad, bad ? "FAIL" : "OK");
        fails += bad != 0;
    }
    {   // test 3
        std::vector<unsigned> S(64 * 4), O(256);
        for (int i = 0; i < 256; ++i) S[i] = 1000 + i;
        unsigned *dS, *dO; CK(hipMalloc(&dS, 1024)); CK(hipMalloc(&dO, 1024));
        CK(hipMemcpy(dS, S.data(), 1024, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(dS, dO); CK(hipDeviceSynchronize());
        CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { unsigned e = 1000 + 4 * (l ^ 5) + j; if (O[l * 4 + j] != e) { if (bad < 8) printf("  glds lane %d w %d got %u expect %u\n", l, j, O[l * 4 + j], e); ++bad; } }
        printf("[global_load_lds x4 lane-linear dest, per-lane src] mismatches=%d %s\n", bad, bad ? "FAIL" : "OK");
        fails += bad != 0;
    }
    {   // test 4
        const int n = 1 << 16;
        std::vector<float> in(n); std::vector<unsigned short> out(n);
        uint32_t s = 12345;
        for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; uint32_t u = (s & 0x807fffffu) | (((s >> 23) % 40 + 107) << 23); memcpy(&in[i], &u, 4); }
        in[0] = 1.00390625f; in[1] = 1.01171875f; in[2] = -1.00390625f; in[3] = 0.f;   // ties
        float* dI; unsigned short* dO; CK(hipMalloc(&dI, n * 4)); CK(hipMalloc(&dO, n * 2));
        CK(hipMemcpy(dI, in.data(), n * 4, hipMemcpyHostToDevice));
        k_cvt<<<n / 256, 256>>>(dI, dO, n); CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dO, n * 2, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < n; ++i) if (out[i] != host_f2bf(in[i])) { if (bad < 4) printf("  cvt %g got %04x expect %04x\n", in[i], out[i], host_f2bf(in[i])); ++bad; }
        printf("[f32->bf16 RNE] mismatches=%d %s\n", bad, bad ? "FAIL" : "OK");
        fails += bad != 0;
        (void)host_bf2f;
    }
    printf("probe: %s\n", fails ? "FAILED" : "ALL OK");
    return fails ? 1 : 0;
}
