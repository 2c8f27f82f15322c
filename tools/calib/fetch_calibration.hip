// Development tool: what FETCH_SIZE (rocprofv3 --pmc) reports on gfx950 for access patterns
// other than a wide coalesced stream, which is the only one the counter is calibrated for
// (there it tallies a 128-byte request as 64 bytes).  Each kernel touches a known number of
// 128-byte lines and 64-byte halves of a buffer larger than the Infinity Cache; the counter
// per launch against those numbers tells the granule of a gather's fetch.
//   mode 0: every lane reads 16 B, consecutive lanes consecutive addresses (a stream)
//   mode 1: every lane reads 16 B at a stride of 128 B (one read per line, first half only)
//   mode 2: every lane reads 16 B at a stride of  64 B (one read per 64-byte half)
//   mode 3: every lane reads 16 B at a stride of  32 B
//   mode 4: every lane reads 12 B at a stride of 128 B (the size of a packed normal)
//   mode 5: every lane reads  4 B at a stride of 128 B
// Build: hipcc --offload-arch=gfx950 -O3 -o build_ab/fetch_calibration tools/calib/fetch_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int BYTES>
__global__ void strided(const char *base, size_t stride, size_t count, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const char *p = base + i * stride;
    unsigned acc;
    if (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(p); acc = v.x ^ v.y ^ v.z ^ v.w; }
    else if (BYTES == 12) { const float *f = reinterpret_cast<const float *>(p); acc = __float_as_uint(f[0]) ^ __float_as_uint(f[1]) ^ __float_as_uint(f[2]); }
    else acc = *reinterpret_cast<const unsigned *>(p);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;           // 1 GiB: four times the Infinity Cache
    char *buf; unsigned *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const size_t loads = (size_t)1 << 22;           // 4 Mi loads per launch in every mode
    struct { int bytes; size_t stride; const char *what; } modes[] = {
        { 16, 16, "stream 16 B/lane" }, { 16, 128, "16 B per 128 B" }, { 16, 64, "16 B per 64 B" },
        { 16, 32, "16 B per 32 B" }, { 12, 128, "12 B per 128 B" }, { 4, 128, "4 B per 128 B" } };
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < 6; ++m) {
            // every launch reads a part of the buffer no earlier launch of this run has read
            const size_t span = loads * modes[m].stride;
            static size_t offset = 0;
            if (offset + span > bytes) offset = 0;
            const dim3 grid((unsigned)(loads / 256)), block(256);
            if (modes[m].bytes == 16) hipLaunchKernelGGL(strided<16>, grid, block, 0, 0, buf + offset, modes[m].stride, loads, sink);
            else if (modes[m].bytes == 12) hipLaunchKernelGGL(strided<12>, grid, block, 0, 0, buf + offset, modes[m].stride, loads, sink);
            else hipLaunchKernelGGL(strided<4>, grid, block, 0, 0, buf + offset, modes[m].stride, loads, sink);
            hipDeviceSynchronize();
            printf("mode %d (%s): %zu loads, %zu bytes spanned, %zu lines of 128 B, %zu halves of 64 B\n", m,
                   modes[m].what, loads, span, (span + 127) / 128, modes[m].stride >= 64 ? loads : (span + 63) / 64);
            offset += span;
        }
    hipFree(buf); hipFree(sink);
    return 0;
}
