// calib_kernels.hip -- known-byte-count kernels in THIS path's access widths, used to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
// pattern before trusting an absolute").  Build: hipcc --offload-arch=gfx950 -O3 tools/calib_kernels.hip -o tools/calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void calib_read_u32(const unsigned *src, unsigned *sink, size_t n_words)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n_words; i += stride) acc ^= src[i];          // one dword per lane, coalesced (the MT19937 staging loads)
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void calib_write_u8_rows(unsigned char *dst, size_t n_rows)
{
    // 180-byte rows written as three byte-per-lane stores by one wave (the legal-mask write-out)
    size_t row = (size_t)blockIdx.x;
    unsigned l = threadIdx.x;
    for (; row < n_rows; row += gridDim.x) {
        unsigned char *p = dst + row * 180;
        p[l] = (unsigned char)l;
        p[l + 64] = (unsigned char)l;
        if (l < 52) p[l + 128] = (unsigned char)l;
    }
}

__global__ void calib_write_u8_rows192(unsigned char *dst, size_t n_rows)
{
    // the two-games-per-wave kernel's mask write-out: rows at a 192-byte pitch, a wave writes TWO adjacent rows with six
    // byte-per-lane stores (lanes 0..31 -> bytes 32 w .. 32 w + 31 of the first row, lanes 32..63 -> of the second)
    size_t pair = (size_t)blockIdx.x;
    unsigned l = threadIdx.x & 31u, half = threadIdx.x >> 5;
    for (; 2 * pair + 1 < n_rows; pair += gridDim.x) {
        unsigned char *p = dst + (2 * pair + half) * 192 + l;
        for (int w = 0; w < 6; w++) p[32 * w] = (unsigned char)l;
    }
}

__global__ void calib_write_u64_rows192(unsigned char *dst, size_t n_rows)
{
    // round 3's mask write-out: rows at a 192-byte pitch, a wave writes TWO adjacent rows with ONE 8-byte store per lane
    // (lanes 0..22 of a half -> bytes 8 j .. 8 j + 7 of the half's row, lanes 23..31 repeat lane 22: 184 bytes per row)
    size_t pair = (size_t)blockIdx.x;
    unsigned l = threadIdx.x & 31u, half = threadIdx.x >> 5;
    unsigned j = l < 22u ? l : 22u;
    for (; 2 * pair + 1 < n_rows; pair += gridDim.x)
        *(unsigned long long *)(dst + (2 * pair + half) * 192 + 8u * j) = 0x0101010101010101ull * j;
}

__global__ void calib_write_u32(unsigned *dst, size_t n_words)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_words; i += stride) dst[i] = (unsigned)i;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;                       // 1 GiB, beyond the 256 MiB Infinity Cache
    unsigned *a, *sink;
    unsigned char *b;
    hipMalloc(&a, bytes); hipMalloc(&sink, 64); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    const size_t rows = bytes / 180;
    for (int rep = 0; rep < 3; rep++) {
        calib_read_u32<<<4096, 256>>>(a, sink, bytes / 4);
        calib_write_u8_rows<<<8192, 64>>>(b, rows);
        calib_write_u8_rows192<<<8192, 64>>>(b, bytes / 192);
        calib_write_u64_rows192<<<8192, 64>>>(b, bytes / 192);
        calib_write_u32<<<4096, 256>>>((unsigned *)b, bytes / 4);
    }
    hipDeviceSynchronize();
    printf("calib: read_u32 bytes=%zu write_u8_rows bytes=%zu write_u8_rows192 bytes=%zu write_u64_rows192 bytes=%zu write_u32 bytes=%zu\n", bytes,
           rows * 180, (bytes / 192 / 2) * 2 * 192, (bytes / 192 / 2) * 2 * 184, bytes);
    return 0;
}
