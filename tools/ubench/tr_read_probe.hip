// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does each lane receive?  LDS holds u16 = its own element index; lane L passes
// the byte address given by a[L].  Build + run:  hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(size_t)lds;      // LDS byte offset of the array (low 32 bits of the generic->local address)
    a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds));
    unsigned byte = a + (unsigned)addr[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: lane L -> row L/4 (row stride 64 elements = 128 B), 4-element column group L%4      (a [16 rows][16 cols] block per 64 lanes?)
        // mode 1: lane L -> row (L%16)/4 + 4*(L/16) ... i.e. per 16-lane group a [4 rows][16 cols] block, groups stacked by rows
        for (int L = 0; L < 64; ++L) {
            int row = mode == 0 ? L / 4 : (L % 16) / 4 + 4 * (L / 16);
            int colg = L % 4;
            h_addr[L] = (row * 64 + colg * 4) * 2;
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d (element index = row*64 + col)\n", mode);
        for (int L = 0; L < 64; ++L) {
            printf("lane %2d addr(row %2d,col %2d):", L, h_addr[L] / 2 / 64, h_addr[L] / 2 % 64);
            for (int j = 0; j < 4; ++j) printf("  (r%2d,c%2d)", h_out[L * 4 + j] / 64, h_out[L * 4 + j] % 64);
            printf("\n");
        }
    }
    return 0;
}
