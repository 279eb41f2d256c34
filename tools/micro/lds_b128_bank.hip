// Microbenchmark: cycles per ds_read_b128 wave-instruction for candidate B-fragment address patterns
// (lane = 16*lh + lr reads 16 B for "position" lr and k-slice lh).  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) bench(const int *__restrict__ offs, int npat, unsigned long long *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    for (int i = threadIdx.x; i < 65536 / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int pat = 0; pat < npat; ++pat) {
        const int a = offs[pat * 64 + lane];
        uint4 acc = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned lbase = (unsigned)(uintptr_t)smem;  // LDS byte address of the buffer
        for (int it = 0; it < iters; ++it) {
            uint4 v0, v1, v2, v3, v4, v5, v6, v7;
            // volatile asm: the loads are loop-invariant and would otherwise be hoisted out of the loop
#define RD(V, U) asm volatile("ds_read_b128 %0, %1" : "=v"(V) : "v"(lbase + ((a + (U) * 4608) & 0xFFFF)))
            RD(v0, 0); RD(v1, 1); RD(v2, 2); RD(v3, 3); RD(v4, 4); RD(v5, 5); RD(v6, 6); RD(v7, 7);
#undef RD
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x ^ v4.x ^ v5.x ^ v6.x ^ v7.x;
        }
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[blockIdx.x * npat + pat] = t1 - t0;
        if (acc.x == 0x12345678 && acc.y == 7) out[0] = acc.z;  // keep the loads alive
    }
}

int main() {
    std::vector<std::vector<int>> pats;
    std::vector<const char *> names;
    auto add = [&](const char *name, auto f) {
        std::vector<int> v(64);
        for (int l = 0; l < 64; ++l) v[l] = f(l & 15, l >> 4);
        pats.push_back(v);
        names.push_back(name);
    };
    const int PLANE = 32256;
    add("linear lane*16", [](int lr, int lh) { return (lh * 16 + lr) * 16; });
    add("v1: p*272 + lh*16", [](int lr, int lh) { return lr * 272 + lh * 16; });
    add("v2: (lh&1)*PLANE + p*144 + (lh>>1)*16", [=](int lr, int lh) { return (lh & 1) * PLANE + lr * 144 + (lh >> 1) * 16; });
    add("p*256 + lh*16 (worst)", [](int lr, int lh) { return lr * 256 + lh * 16; });
    add("4 planes: lh*PLANE/2 + p*80", [=](int lr, int lh) { return lh * 16128 + lr * 80; });
    add("p*272 + lh*64", [](int lr, int lh) { return lr * 272 + lh * 64; });
    add("p*288 + lh*16", [](int lr, int lh) { return lr * 288 + lh * 16; });
    add("p*144 + lh*16 (one plane, 64B data)", [](int lr, int lh) { return lr * 144 + lh * 16; });
    add("p*528 + lh*16", [](int lr, int lh) { return lr * 528 + lh * 16; });
    add("xor: p*256 + ((lh ^ p)&15)*16", [](int lr, int lh) { return lr * 256 + ((lh ^ lr) & 15) * 16; });
    add("p*64 + lh*16 (dense 64B rows)", [](int lr, int lh) { return lr * 64 + lh * 16; });
    add("v2 with lr in odd order", [=](int lr, int lh) { int p = (lr * 5) & 15; return (lh & 1) * PLANE + p * 144 + (lh >> 1) * 16; });
    const int np = (int)pats.size();
    std::vector<int> flat;
    for (auto &v : pats) flat.insert(flat.end(), v.begin(), v.end());
    int *d_offs;
    unsigned long long *d_out;
    hipMalloc(&d_offs, flat.size() * 4);
    hipMalloc(&d_out, 8 * np * 8);
    hipMemcpy(d_offs, flat.data(), flat.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench, dim3(1), dim3(256), 0, 0, d_offs, np, d_out, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> out(np);
    hipMemcpy(out.data(), d_out, np * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < np; ++i)
        printf("%-45s %6.2f cycles per ds_read_b128 per wave (4 waves on the CU: x4 per CU)\n", names[i], (double)out[i] / (iters * 8.0));
    return 0;
}
