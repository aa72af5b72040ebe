// Timing probe for the attention kernels: includes the library sources directly so that experiment macros can switch parts off.
// Build: hipcc --offload-arch=gfx950 -O3 -I../../include [-DATTN_PROBE_x] attn_probe.hip -o attn_probe
#include "../../vision_transformers_torch_xla_amd/csrc/api.hip"
#include "../../vision_transformers_torch_xla_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)
int main(int argc, char** argv) {
    const int B = 512, N = 197, H = 12, D = 64 * H;
    const size_t nq = (size_t)B * N * 3 * D, no = (size_t)B * N * D;
    std::vector<unsigned short> h(nq);
    unsigned s = 12345;
    for (size_t i = 0; i < nq; ++i) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.0f * 3.0f - 1.5f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    unsigned short *qkv, *out, *dout, *dqkv; float* lse;
    CK(hipMalloc(&qkv, nq * 2)); CK(hipMalloc(&out, no * 2)); CK(hipMalloc(&dout, no * 2)); CK(hipMalloc(&dqkv, nq * 2)); CK(hipMalloc(&lse, (size_t)B * H * N * 4));
    CK(hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dout, h.data(), no * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool bwd = argc > 1 && argv[1][0] == 'b';
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) {
            int rc = bwd ? vitk_attention_bwd(qkv, out, dout, lse, dqkv, B, N, H, 0.125f, nullptr, 0, nullptr)
                         : vitk_attention_fwd(qkv, out, lse, B, N, H, 0.125f, nullptr);
            if (rc) { printf("error %d: %s\n", rc, vitk_last_error()); return 1; }
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("%s %s: %.1f us\n", bwd ? "bwd" : "fwd", argc > 2 ? argv[2] : "", ms * 100.0f);
    }
    return 0;
}
