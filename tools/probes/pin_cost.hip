// What does pinned staging memory cost to get?  hipHostMalloc against malloc + touch (1 / 8 threads, with and without transparent huge
// pages) + hipHostRegister, for the sizes the host path's slots use.   hipcc -O2 --offload-arch=gfx950 pin_cost.hip -o pin_cost -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(uint8_t* p, size_t n, int ways) {
    std::vector<std::thread> th;
    const size_t piece = (n / ways + 4095) & ~(size_t)4095;
    for (int w = 0; w < ways; ++w)
        th.emplace_back([=] { for (size_t k = (size_t)w * piece; k < std::min(n, (size_t)(w + 1) * piece); k += 4096) p[k] = 1; });
    for (auto& t : th) t.join();
}
int main() {
    (void)hipFree(nullptr);
    void* d = nullptr;
    (void)hipMalloc(&d, (size_t)256 << 20);
    for (size_t mb : {36, 72, 256}) {
        const size_t n = mb << 20;
        double t0 = now();
        void* p = nullptr;
        if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return 1;
        double t1 = now();
        std::printf("%4zu MiB  hipHostMalloc %.1f ms", mb, (t1 - t0) * 1e3);
        t0 = now(); (void)hipMemcpy(d, p, n, hipMemcpyHostToDevice); t1 = now();
        std::printf("  (first H2D from it %.1f ms)", (t1 - t0) * 1e3);
        t0 = now(); (void)hipHostFree(p); t1 = now();
        std::printf("  free %.1f ms\n", (t1 - t0) * 1e3);
        for (int huge = 0; huge < 2; ++huge)
            for (int ways : {1, 8}) {
                t0 = now();
                void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                if (huge) madvise(m, n, MADV_HUGEPAGE);
                touch(static_cast<uint8_t*>(m), n, ways);
                t1 = now();
                hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
                double t2 = now();
                (void)hipMemcpy(d, m, n, hipMemcpyHostToDevice);
                double t3 = now();
                (void)hipMemcpy(d, m, n, hipMemcpyHostToDevice);
                double t4 = now();
                (void)hipHostUnregister(m);
                munmap(m, n);
                std::printf("          mmap%s + touch x%d %.1f ms, hipHostRegister %.1f ms (%s), H2D %.1f / %.1f ms\n", huge ? " (THP)" : "", ways, (t1 - t0) * 1e3,
                            (t2 - t1) * 1e3, e == hipSuccess ? "ok" : "FAILED", (t3 - t2) * 1e3, (t4 - t3) * 1e3);
            }
    }
    return 0;
}
