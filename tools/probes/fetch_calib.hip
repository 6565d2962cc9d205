// Calibration of the PMC counters FETCH_SIZE / WRITE_SIZE in the access patterns the scan kernels use (bench.py doubled
// FETCH_SIZE for every kernel, while per-lane 16/64-byte requests are counted exactly).  Every kernel below moves a KNOWN number of bytes
// (1 GiB) in one pattern; run under `rocprofv3 --pmc FETCH_SIZE` (and `--pmc WRITE_SIZE`) the per-kernel counter against that number is
// the factor bench.py applies to the kernels that read or write that way (profiles/r05_fetch_calibration.txt).
//   fetch_calib            runs every pattern once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// reads -------------------------------------------------------------------------------------------------------------------------------
// wide coalesced: lane i of a wave reads 16 bytes at base + 16 i (k_bytemap, the splice's input requests, the probe kernels)
__global__ void cal_read_wide16(const uint8_t* in, int64_t n, uint32_t* sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 16;
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; v < n; v += stride) acc ^= *reinterpret_cast<const u32x4*>(in + v);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
// per lane, 16 bytes at a time, lanes `lane_bytes` apart (k_fb_mark4, k_bt, the single-block walkers)
__global__ void cal_read_lane16(const uint8_t* in, int64_t n, int64_t lane_bytes, uint32_t* sink) {
    const int64_t lo = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * lane_bytes;
    u32x4 acc = {0, 0, 0, 0};
    if (lo + lane_bytes <= n)
        for (int64_t k = 0; k < lane_bytes; k += 16) acc ^= *reinterpret_cast<const u32x4*>(in + lo + k);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
// per lane, 64 bytes at a time as four 16-byte loads issued together (g16_lane, fb_lane, rev_sweep_lane: a piece)
__global__ void cal_read_lane64(const uint8_t* in, int64_t n, int64_t lane_bytes, uint32_t* sink) {
    const int64_t lo = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * lane_bytes;
    u32x4 acc = {0, 0, 0, 0};
    if (lo + lane_bytes <= n)
        for (int64_t k = 0; k < lane_bytes; k += 64) {
            const u32x4* p = reinterpret_cast<const u32x4*>(in + lo + k);
            const u32x4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc ^= a ^ b ^ c ^ d;
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
// straight to LDS, four adjacent lanes per 64-byte row, rows `lane_bytes` apart (k_stream_lpw's tiles)
__device__ __forceinline__ void glds16(const uint8_t* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void cal_read_lds_rows64(const uint8_t* in, int64_t n, int64_t lane_bytes, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t* tile = smem + wv * 4096;
    const uint32_t t0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*)tile));
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv, wave_lo = wave * 64 * lane_bytes;
    if (wave_lo + 64 * lane_bytes > n) return;
    uint32_t acc = 0;
    for (int64_t k = 0; k < lane_bytes; k += 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(in + wave_lo + (16 * i + lid / 4) * lane_bytes + 16 * (lid % 4) + k, t0 + i * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= *reinterpret_cast<const uint32_t*>(tile + lid * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678u) *sink = 1;
}
// writes ------------------------------------------------------------------------------------------------------------------------------
__global__ void cal_write_wide16(uint8_t* out, int64_t n) {                        // k_bytemap, the splice's tile stores
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 16;
    const u32x4 v = {1, 2, 3, 4};
    for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; p < n; p += stride) *reinterpret_cast<u32x4*>(out + p) = v;
}
__global__ void cal_write_unit64(uint8_t* out, int64_t n, int64_t lane_bytes) {    // four adjacent lanes store one lane's 64-byte unit (the emit pass's unit stores)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t lane = t >> 2, lo = lane * lane_bytes;
    const u32x4 v = {1, 2, 3, 4};
    if (lo + lane_bytes <= n)
        for (int64_t k = 0; k < lane_bytes; k += 64) *reinterpret_cast<u32x4*>(out + lo + k + 16 * (t & 3)) = v;
}
__global__ void cal_write_lane16(uint8_t* out, int64_t n, int64_t lane_bytes) {    // a lane's own 16-byte stores, lanes `lane_bytes` apart (k_rev_sweep's edge waves, the LP emit)
    const int64_t lo = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * lane_bytes;
    const u32x4 v = {1, 2, 3, 4};
    if (lo + lane_bytes <= n)
        for (int64_t k = 0; k < lane_bytes; k += 16) *reinterpret_cast<u32x4*>(out + lo + k) = v;
}
__global__ void cal_write_lane4(uint8_t* out, int64_t n, int64_t lane_bytes) {     // a lane's own 4-byte stores (the mark pass's events)
    const int64_t lo = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * lane_bytes;
    if (lo + lane_bytes <= n)
        for (int64_t k = 0; k < lane_bytes; k += 4) *reinterpret_cast<uint32_t*>(out + lo + k) = (uint32_t)k;
}

int main() {
    const int64_t n = 1ll << 30;
    uint8_t *in, *out;
    uint32_t* sink;
    (void)hipMalloc(&in, n); (void)hipMalloc(&out, n); (void)hipMalloc(&sink, 4);
    (void)hipMemset(in, 1, n); (void)hipMemset(out, 0, n);
    (void)hipDeviceSynchronize();
    const int64_t lb = 4096;                          // sub-range per lane (the walkers use 2-16 KiB)
    const int lanes = (int)(n / lb);
    hipLaunchKernelGGL(cal_read_wide16, dim3(8192), dim3(256), 0, 0, in, n, sink);
    hipLaunchKernelGGL(cal_read_lane16, dim3(lanes / 256), dim3(256), 0, 0, in, n, lb, sink);
    hipLaunchKernelGGL(cal_read_lane64, dim3(lanes / 256), dim3(256), 0, 0, in, n, lb, sink);
    hipLaunchKernelGGL(cal_read_lds_rows64, dim3(lanes / 256), dim3(256), 4 * 4096, 0, in, n, lb, sink);
    hipLaunchKernelGGL(cal_write_wide16, dim3(8192), dim3(256), 0, 0, out, n);
    hipLaunchKernelGGL(cal_write_unit64, dim3(lanes * 4 / 256), dim3(256), 0, 0, out, n, lb);
    hipLaunchKernelGGL(cal_write_lane16, dim3(lanes / 256), dim3(256), 0, 0, out, n, lb);
    hipLaunchKernelGGL(cal_write_lane4, dim3(lanes / 256), dim3(256), 0, 0, out, n, lb);
    (void)hipDeviceSynchronize();
    printf("fetch_calib: every kernel moved %lld bytes (%lld KB)\n", (long long)n, (long long)(n >> 10));
    return 0;
}
