// Probe: how fast can a wave move 64 strided rows of ROWB bytes through an LDS tile (direct-to-LDS
// loads, row-transposed stores), as the wave-tiled window kernel does -- without any compute.
//   io_probe <rowb: 64|128|256> <lane_bytes> <waves_per_wg> <mode: 0 load+store, 1 load only, 2 store only>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const uint8_t* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16nt(const uint8_t* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int ROWB>
__global__ void k_io(const uint8_t* in, uint8_t* out, int64_t n, int64_t lane_bytes, int mode_all) {
    const int mode = mode_all & 3; const bool nts = mode_all & 4, ntl = mode_all & 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int LPR = ROWB / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;           // rows per instruction
    constexpr int NI = 64 / RPI;            // instructions per tile
    constexpr int TILE = 64 * ROWB;
    const int lid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t* tile = smem + wv * TILE;
    const uint32_t t0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*)tile));
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const int64_t wave_lo = wave * 64 * lane_bytes;
    if (wave_lo + 64 * lane_bytes > n) return;
    int32_t src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) src[i] = (int32_t)((RPI * i + lid / LPR) * lane_bytes) + 16 * (lid % LPR);
    const uint8_t* wi = in + wave_lo;
    uint8_t* wo = out + wave_lo;
    if (mode != 2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) { if (ntl) glds16nt(wi + src[i], t0 + i * 1024); else glds16(wi + src[i], t0 + i * 1024); }
    }
    for (int32_t k = 0; k < lane_bytes; k += ROWB) {
        if (mode == 0 && k != 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI) : "memory");   // leave the previous tile's stores in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4 v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const u32x4*>(tile + i * 1024 + lid * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mode != 2 && k + ROWB < lane_bytes) {
#pragma unroll
            for (int i = 0; i < NI; ++i) { if (ntl) glds16nt(wi + src[i] + k + ROWB, t0 + i * 1024); else glds16(wi + src[i] + k + ROWB, t0 + i * 1024); }
        }
        if (mode != 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i) { if (nts) __builtin_nontemporal_store(v[i], reinterpret_cast<u32x4*>(wo + src[i] + k)); else *reinterpret_cast<u32x4*>(wo + src[i] + k) = v[i]; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// mixed: 64-byte input rows every iteration, 128-byte output rows every second iteration
__global__ void k_io_mixed(const uint8_t* in, uint8_t* out, int64_t n, int64_t lane_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t* tin = smem + wv * 12288;
    uint8_t* tout = tin + 4096;
    const uint32_t t0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*)tin));
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const int64_t wave_lo = wave * 64 * lane_bytes;
    if (wave_lo + 64 * lane_bytes > n) return;
    int32_t src[4], dst[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) src[i] = (int32_t)((16 * i + lid / 4) * lane_bytes) + 16 * (lid % 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = (int32_t)((8 * i + lid / 8) * lane_bytes) + 16 * (lid % 8);
    const uint8_t* wi = in + wave_lo;
    uint8_t* wo = out + wave_lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(wi + src[i], t0 + i * 1024);
    bool stored = false;
    for (int32_t k = 0; k < lane_bytes; k += 64) {
        if (stored) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(tin + i * 1024 + lid * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (k + 64 < lane_bytes) {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(wi + src[i] + k + 64, t0 + i * 1024);
        }
        // each lane puts its 64 bytes into its 128-byte output row (half k/64 & 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tout + lid * 128 + ((k >> 6) & 1) * 64 + i * 16) = v[i];
        stored = false;
        if ((k >> 6) & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(tout + i * 1024 + lid * 16);
                *reinterpret_cast<u32x4*>(wo + dst[i] + k - 64) = o;
            }
            stored = true;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
int main(int argc, char** argv) {
    const int rowb = argc > 1 ? atoi(argv[1]) : 64;
    const int64_t lane_bytes = argc > 2 ? atoll(argv[2]) : 2048;
    const int waves = argc > 3 ? atoi(argv[3]) : 4;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int64_t n = 1ll << 30;
    uint8_t *in, *out;
    (void)hipMalloc(&in, n); (void)hipMalloc(&out, n);
    (void)hipMemset(in, 1, n); (void)hipMemset(out, 0, n);
    const int64_t n_waves = n / (64 * lane_bytes);
    const int grid = (int)((n_waves + waves - 1) / waves);
    const int lds = waves * 64 * rowb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        if (rowb == 0) hipLaunchKernelGGL(k_io_mixed, dim3(grid), dim3(waves * 64), waves * 12288, 0, in, out, n, lane_bytes);
        else if (rowb == 64) hipLaunchKernelGGL(k_io<64>, dim3(grid), dim3(waves * 64), lds, 0, in, out, n, lane_bytes, mode);
        else if (rowb == 128) hipLaunchKernelGGL(k_io<128>, dim3(grid), dim3(waves * 64), lds, 0, in, out, n, lane_bytes, mode);
        else hipLaunchKernelGGL(k_io<256>, dim3(grid), dim3(waves * 64), lds, 0, in, out, n, lane_bytes, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("rowb=%d lane_bytes=%lld waves/wg=%d lds/wg=%d mode=%d: %.3f ms  %.0f GB/s input\n", rowb, (long long)lane_bytes, waves, lds, mode, best, n / best / 1e6);
    return 0;
}
