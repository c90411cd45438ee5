// LDS-DMA semantics probe for gfx950: `buffer_load_dwordx4 ... lds` (16 bytes per lane).
//   * where do lanes land (wave-uniform base + lane * 16)?
//   * what does an out-of-range lane (buffer range check) leave in LDS: zeros or the old bytes?
// Build: python build.py (benchmarks/native/*.hip); run on an MI355X: benchmarks/native/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const unsigned* src, unsigned src_bytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned smem[2 * 256];  // 2 waves x 1 KiB
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 512; i += blockDim.x) smem[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
    // lanes 0..47 read element block (63 - lane) of their wave's 1 KiB source window (a permutation:
    // shows the per-lane SOURCE address is free), lanes 48..63 are out of range
    const int lane = tid & 63;
    unsigned voff = lane < 48 ? (unsigned)(wave * 1024 + (63 - lane) * 16) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 256), 16, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 512; i += blockDim.x) out[i] = smem[i];
}

int main() {
    std::vector<unsigned> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 1000 + i;
    unsigned *d_src, *d_out;
    hipMalloc(&d_src, 2048); hipMalloc(&d_out, 2048);
    hipMemcpy(d_src, h.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d_src, 2048u, d_out);
    std::vector<unsigned> o(512);
    if (hipMemcpy(o.data(), d_out, 2048, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    int ok_perm = 1, oob_zero = 1, oob_old = 1;
    for (int w = 0; w < 2; ++w)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
                unsigned got = o[w * 256 + lane * 4 + e];
                if (lane < 48) { if (got != 1000u + w * 256 + (63 - lane) * 4 + e) ok_perm = 0; }
                else { if (got != 0u) oob_zero = 0; if (got != 0xdeadbeefu) oob_old = 0; }
            }
    printf("lane-linear destination with per-lane source: %s\n", ok_perm ? "yes" : "NO");
    printf("out-of-range lanes: %s (sample 0x%08x)\n", oob_zero ? "write zeros" : oob_old ? "leave LDS untouched" : "something else", o[48 * 4]);
    return ok_perm ? 0 : 2;
}
