// Semantics probe for global_load_lds_dwordx4 on gfx950: which LDS bytes does lane i write, and do EXEC-masked lanes write?
// Expected (and relied on by csrc/mapf_wgrad.hip): LDS address = M0 base + 16 * lane id, global address per lane arbitrary,
// inactive lanes write nothing.   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_direct_load.hip -o /tmp/ldl && /tmp/ldl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint32_t *g, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t smem[36864];  // 144 KB: the target chunk sits above 64 KB (is M0 wider than 16 bits?)
    const int lane = threadIdx.x;
    for (int i = lane; i < 36864; i += 64) smem[i] = 0xFFFFFFFFu;
    __syncthreads();
    typedef __attribute__((address_space(3))) void *lptr;
    typedef const __attribute__((address_space(1))) void *gptr;
    constexpr int HI = 32768;  // words: 128 KB into the array; lanes read reversed 16-byte chunks
    // the form csrc/mapf_wgrad.hip uses: inline asm, so that hipcc's wait-count pass does not serialise later LDS reads on it
    const uint32_t lds_off = (uint32_t)(uintptr_t)(lptr)(smem + HI + 256);
    const uint32_t voff = (uint32_t)((63 - lane) * 16);
    if (lane < 10 || lane >= 20)
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_off), "v"(voff), "s"(g) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = smem[HI + i];
}
int main() {
    std::vector<uint32_t> h(256);
    for (int i = 0; i < 256; ++i) h[i] = i;
    uint32_t *g, *o;
    hipMalloc(&g, 1024);
    hipMalloc(&o, 4096);
    hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    std::vector<uint32_t> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
        uint32_t want = 0xFFFFFFFFu;
        if (i >= 256 && i < 512) {
            const int lane = (i - 256) / 4;
            if (lane < 10 || lane >= 20) want = (63 - lane) * 4 + (i & 3);
        }
        if (r[i] != want && bad++ < 8) printf("word %d: got %u want %u\n", i, r[i], want);
    }
    printf(bad ? "MISMATCH %d\n" : "ok: lane i writes LDS base + 16 i; masked lanes write nothing\n", bad);
    return bad != 0;
}
