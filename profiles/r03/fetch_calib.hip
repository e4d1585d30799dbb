// FETCH_SIZE calibration for gfx950 (VERDICT r02 "next" 2a): MI355X_MICROARCH.md calibrates rocprofv3's FETCH_SIZE only for
// 16-byte-per-lane streaming reads (it reports half of the bytes) and says to calibrate other access patterns on a known byte
// count.  The LK kernel stages its windows with 4-byte lane loads of short row segments, so this program reads KNOWN byte
// counts in those patterns; run it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare per kernel:
//   read16      dense stream, 16 B per lane                       bytes = N
//   read4       dense stream, 4 B per lane                        bytes = N
//   read1       dense stream, 1 B per lane                        bytes = N / 4 (quarter of the buffer)
//   sparse128   one dword of every 128-byte line                  lines touched = N / 128
//   sparse64    one dword of every 64-byte half line              half lines touched = N / 64
//   rows32      LK-like: a 16-lane group reads 24 rows x 32 B (8 dwords) at a random unaligned-to-128 window of a 768-pitch image
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; the program prints the byte counts and wall-clock GB/s
// (the time bounds the request size: N/128 lines per second x 128 B cannot exceed the HBM rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read16(const uint4* __restrict__ p, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void read4(const unsigned* __restrict__ p, size_t n4, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345u) *sink = acc;
}
__global__ void read1(const unsigned char* __restrict__ p, size_t n1, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345u) *sink = acc;
}
__global__ void sparse(const unsigned* __restrict__ p, size_t nseg, int seg_dwords, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nseg; i += (size_t)gridDim.x * blockDim.x) acc += p[i * seg_dwords];
    if (acc == 0x12345u) *sink = acc;
}
// one 16-lane group per window: lane l reads dword (l & 7) of rows (l >> 3) + 2 k, k = 0..11 (24 rows x 32 B)
__global__ void rows32(const unsigned char* __restrict__ img, const unsigned* __restrict__ win_off, size_t nwin, int pitch, unsigned* sink) {
    unsigned acc = 0;
    size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int l = threadIdx.x & 15;
    if (g < nwin) {
        const unsigned char* base = img + win_off[g];
        for (int k = 0; k < 12; ++k) {
            const unsigned char* q = base + (size_t)((l >> 3) + 2 * k) * pitch + 4 * (l & 7);
            unsigned v; __builtin_memcpy(&v, q, 4);
            acc += v;
        }
    }
    if (acc == 0x12345u) *sink = acc;
}

int main(int argc, char** argv) {
    size_t N = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30);
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, N)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, N));
    // windows: images of 768 x 512 bytes tiled over the buffer, 900 windows per image at pseudo-random offsets
    const int pitch = 768, rows = 512; size_t nimg = N / ((size_t)pitch * rows), per = 900, nwin = nimg * per;
    std::vector<unsigned> off(nwin);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < nwin; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        unsigned x = (unsigned)(s % (pitch - 36)), y = (unsigned)((s >> 32) % (rows - 26));
        size_t o = (i / per) * (size_t)pitch * rows + (size_t)y * pitch + x;
        off[i] = (unsigned)o;                                         // N <= 4 GiB keeps this in 32 bits
    }
    unsigned* d_off; CK(hipMalloc(&d_off, nwin * 4)); CK(hipMemcpy(d_off, off.data(), nwin * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, double bytes, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("{\"kernel\": \"%s\", \"known_bytes\": %.0f, \"ms\": %.4f, \"known_GBps\": %.1f}\n", name, bytes, ms, bytes / ms * 1e-6);
    };
    const int G = 256 * 16, B = 256;
    timed("read16", (double)N, [&] { read16<<<G, B>>>((const uint4*)buf, N / 16, sink); });
    timed("read4", (double)N, [&] { read4<<<G, B>>>((const unsigned*)buf, N / 4, sink); });
    timed("read1", (double)N / 4, [&] { read1<<<G, B>>>(buf, N / 4, sink); });
    timed("sparse128", (double)N / 128 * 4, [&] { sparse<<<G, B>>>((const unsigned*)buf, N / 128, 32, sink); });
    timed("sparse64", (double)N / 64 * 4, [&] { sparse<<<G, B>>>((const unsigned*)buf, N / 64, 16, sink); });
    timed("rows32", (double)nwin * 24 * 32, [&] { rows32<<<(unsigned)((nwin * 16 + B - 1) / B), B>>>(buf, d_off, nwin, pitch, sink); });
    printf("{\"buffer_bytes\": %.0f, \"images\": %zu, \"windows\": %zu}\n", (double)N, nimg, nwin);
    return 0;
}
