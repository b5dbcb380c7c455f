// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the gather-scatter kernel
// (MI355X_MICROARCH.md, section HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel below moves a KNOWN number of bytes from / to arrays far larger than the 256 MiB Infinity Cache:
//   cal_read16   16 B per lane, coalesced                 (the case the guide documents: FETCH_SIZE = bytes / 2)
//   cal_read8     8 B per lane, coalesced
//   cal_pairs8    k_gs's scalar pair path: 8-B loads of a[i], a[j] at indices from an index array, runs of 36 consecutive
//                 doubles (a face interior at lx1 = 8) at scattered run bases, both copies rewritten with the sum
//   cal_pairs16   k_gs's double2 pair path: the same runs, 16-B accesses
//   cal_write8    8 B per lane stores, coalesced
// build + run:  python scripts/pmc_calibrate.py   (hipcc, then two rocprofv3 --pmc passes, then the ratios)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void cal_read16(const double2 *a, int64_t n2, double *out) {
    double s = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 123.456) out[0] = s;
}
__global__ void cal_read8(const double *a, int64_t n, double *out) {
    double s = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456) out[0] = s;
}
__global__ void cal_write8(double *a, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = 1.0;
}
// one pair per thread, 8-byte accesses (k_gs, scalar path)
__global__ void cal_pairs8(const int2 *idx, int64_t npairs, double *f) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= npairs) return;
    const int2 ab = idx[t];
    const double s = f[ab.x] + f[ab.y];
    f[ab.x] = s;
    f[ab.y] = s;
}
// two pairs per thread, 16-byte accesses (k_gs, double2 path)
__global__ void cal_pairs16(const int4 *idx, int64_t npairs2, double *f) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= npairs2) return;
    const int4 q = idx[t];
    const double2 a = *reinterpret_cast<const double2 *>(f + q.x), b = *reinterpret_cast<const double2 *>(f + q.y);
    double2 s;
    s.x = a.x + b.x;
    s.y = a.y + b.y;
    *reinterpret_cast<double2 *>(f + q.x) = s;
    *reinterpret_cast<double2 *>(f + q.y) = s;
}

int main() {
    const int64_t n = (int64_t)1 << 27;   // 128 Mi doubles = 1 GiB per array
    double *a = nullptr, *out = nullptr;
    CK(hipMalloc(&a, sizeof(double) * n));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, sizeof(double) * n));
    // pair lists: runs of 36 doubles; run r of side A starts at 512 * (2 r) + 136, its partner at 512 * (2 r + 1) + 424 (element
    // stride 512 doubles, face blocks inside an element as in the face-grouped layout: never line-aligned)
    const int64_t nrun = n / 1024;
    std::vector<int> idx;
    idx.reserve((size_t)nrun * 72);
    for (int64_t r = 0; r < nrun; ++r)
        for (int q = 0; q < 36; ++q) {
            idx.push_back((int)(1024 * r + 136 + q));
            idx.push_back((int)(1024 * r + 512 + 424 + q));
        }
    const int64_t npairs = nrun * 36;
    int *d_idx = nullptr, *d_idx4 = nullptr;
    CK(hipMalloc(&d_idx, sizeof(int) * idx.size()));
    CK(hipMemcpy(d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    // the same pairs two at a time: (a, b, a + 1, b + 1) -> the kernel uses (a, b) and 16-byte accesses; runs start at even offsets
    std::vector<int> idx4;
    idx4.reserve((size_t)nrun * 72);
    for (int64_t r = 0; r < nrun; ++r)
        for (int q = 0; q < 36; q += 2) {
            idx4.push_back((int)(1024 * r + 136 + q));
            idx4.push_back((int)(1024 * r + 512 + 424 + q));
            idx4.push_back((int)(1024 * r + 136 + q + 1));
            idx4.push_back((int)(1024 * r + 512 + 424 + q + 1));
        }
    CK(hipMalloc(&d_idx4, sizeof(int) * idx4.size()));
    CK(hipMemcpy(d_idx4, idx4.data(), sizeof(int) * idx4.size(), hipMemcpyHostToDevice));
    const int nt = 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(cal_read16, dim3(8192), dim3(nt), 0, 0, reinterpret_cast<const double2 *>(a), n / 2, out);
        hipLaunchKernelGGL(cal_read8, dim3(8192), dim3(nt), 0, 0, (const double *)a, n, out);
        hipLaunchKernelGGL(cal_write8, dim3(8192), dim3(nt), 0, 0, a, n);
        hipLaunchKernelGGL(cal_pairs8, dim3((unsigned)((npairs + nt - 1) / nt)), dim3(nt), 0, 0, reinterpret_cast<const int2 *>(d_idx), npairs, a);
        hipLaunchKernelGGL(cal_pairs16, dim3((unsigned)((npairs / 2 + nt - 1) / nt)), dim3(nt), 0, 0, reinterpret_cast<const int4 *>(d_idx4), npairs / 2, a);
    }
    CK(hipDeviceSynchronize());
    // known bytes per dispatch
    printf("KNOWN cal_read16 read %lld write 0\n", (long long)(8 * n));
    printf("KNOWN cal_read8 read %lld write 0\n", (long long)(8 * n));
    printf("KNOWN cal_write8 read 0 write %lld\n", (long long)(8 * n));
    // pairs: data 2 x 8 B read + 2 x 8 B written per pair, index 8 B per pair; by whole 128-byte lines touched: a run of 36 doubles
    // at byte offset 1088 (= 136 * 8) spans lines 8 .. 10 (3 lines), at 3392 (= 424 * 8) lines 26 .. 28 (3 lines): 6 lines = 768 B per pair of runs
    printf("KNOWN cal_pairs8 read %lld write %lld lines_read %lld\n", (long long)(npairs * 24), (long long)(npairs * 16), (long long)(nrun * 768 + npairs * 8));
    printf("KNOWN cal_pairs16 read %lld write %lld lines_read %lld\n", (long long)(npairs * 24), (long long)(npairs * 16), (long long)(nrun * 768 + npairs * 8));
    return 0;
}
