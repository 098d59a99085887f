// osc_scan.hip -- float64 running sums (FM with an arbitrary modulator: the rows of a bank's modulation matrix) and the
// elementwise kernel behind the oscillator filters.
#include "osc_host.hpp"

namespace {

// ---- elementwise filters over float64 blocks ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_ew_f64(int op, const double* a, const double* b, size_t n, double p0, double p1,
                                                double* out64, float* out32) {
    const size_t i = sh::block_id() * 256 + threadIdx.x;
    if (i >= n) return;
    double v;
    switch (op) {
    case SH_EW_ADD: v = a[i] + b[i]; break;
    case SH_EW_MUL: v = a[i] * b[i]; break;
    case SH_EW_CLIP: { double t = a[i] < p1 ? a[i] : p1; v = t > p0 ? t : p0; } break;    // max(min(v, maximum), minimum)
    case SH_EW_ABS: v = fabs(a[i]); break;
    case SH_EW_COPY: v = a[i]; break;
    case SH_EW_AXPY: { const double e = b[i] * p0; v = a[i] + e; } break;
    case SH_EW_NEXTUP: v = nextafter(a[i], INFINITY); break;
    default: v = p0; break;
    }
    if (out64) out64[i] = v;
    if (out32) out32[i] = (float)v;
}

// ---- float64 exclusive scan (FM with an arbitrary modulator) ---------------------------
constexpr int SCAN_TILE = 2048;   // values per block (256 threads x 8)

__device__ __forceinline__ double block_exclusive_scan_256(double x, double* sh, double& total) {
    // sh: 256 doubles.  Hillis-Steele; plenty fast for the few MB a modulator block has.
    const int t = threadIdx.x;
    sh[t] = x;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        double add = (t >= o) ? sh[t - o] : 0.0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    total = sh[255];
    double incl = sh[t];
    __syncthreads();
    return incl - x;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(const double* __restrict__ x, uint32_t n, double* __restrict__ sums) {
    __shared__ double sh[256];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (base + j < n) s += x[base + j];
    double total;
    block_exclusive_scan_256(s, sh, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_sums(double* __restrict__ sums, uint32_t ntiles, double carry_in) {
    // single block: exclusive scan of the tile sums in place; sums[ntiles] = grand total
    __shared__ double sh[256];
    __shared__ double carry;
    if (threadIdx.x == 0) carry = carry_in;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 256) {
        uint32_t i = base + threadIdx.x;
        double v = (i < ntiles) ? sums[i] : 0.0;
        double total;
        double ex = block_exclusive_scan_256(v, sh, total);
        double c = carry;
        if (i < ntiles) sums[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[ntiles] = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const double* __restrict__ x, uint32_t n,
                                                    const double* __restrict__ sums, double* __restrict__ out) {
    __shared__ double sh[256];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double v[8];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = (base + j < n) ? x[base + j] : 0.0;
        s += v[j];
    }
    double total;
    double ex = block_exclusive_scan_256(s, sh, total) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
}

// The same scan for many rows at once (the modulators of a bank): blockIdx.y = row; x and out may be the same buffer;
// carry[row] is read as the row's carry-in and replaced by its carry-out, on the device: consecutive blocks chain without a
// host round trip.
__global__ __launch_bounds__(256) void k_scan_rows_tile_sums(const double* __restrict__ x, size_t stride, uint32_t n, uint32_t ntiles,
                                                             double* __restrict__ sums) {
    __shared__ double sh[256];
    const double* xr = x + (size_t)blockIdx.y * stride;
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (base + j < n) s += xr[base + j];
    double total;
    block_exclusive_scan_256(s, sh, total);
    if (threadIdx.x == 0) sums[(size_t)blockIdx.y * (ntiles + 1) + blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_rows_sums(double* __restrict__ sums, uint32_t ntiles, double* __restrict__ carry) {
    __shared__ double sh[256];
    __shared__ double run;
    double* sr = sums + (size_t)blockIdx.x * (ntiles + 1);
    if (threadIdx.x == 0) run = carry[blockIdx.x];
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const double v = (i < ntiles) ? sr[i] : 0.0;
        double total;
        const double ex = block_exclusive_scan_256(v, sh, total);
        const double c = run;
        if (i < ntiles) sr[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) run = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) carry[blockIdx.x] = run;
}

__global__ __launch_bounds__(256) void k_scan_rows_apply(const double* x, size_t stride, uint32_t n, uint32_t ntiles,
                                                         const double* __restrict__ sums, double* out) {
    __shared__ double sh[256];
    const double* xr = x + (size_t)blockIdx.y * stride;
    double* outr = out + (size_t)blockIdx.y * stride;
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    double v[8];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = (base + j < n) ? xr[base + j] : 0.0;
        s += v[j];
    }
    double total;
    double ex = block_exclusive_scan_256(s, sh, total) + sums[(size_t)blockIdx.y * (ntiles + 1) + blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) outr[base + j] = ex;
        ex += v[j];
    }
}

}  // namespace

extern "C" {

int sh_ew_f64(int op, const sh_buf* a, size_t a_off, const sh_buf* b, size_t b_off, size_t n, double p0, double p1,
              sh_buf* out_f64, size_t out64_off, sh_buf* out_f32, size_t out32_off, float* out_host) {
    SH_REQUIRE_INIT();
    if (op < SH_EW_ADD || op > SH_EW_NEXTUP) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: unknown op %d", op);
    const bool need_a = op != SH_EW_FILL, need_b = op == SH_EW_ADD || op == SH_EW_MUL || op == SH_EW_AXPY;
    if (need_a && (!a || a_off > a->bytes / 8 || n > a->bytes / 8 - a_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: operand a too small");
    if (need_b && (!b || b_off > b->bytes / 8 || n > b->bytes / 8 - b_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: operand b too small");
    if (out_f64 && (out64_off > out_f64->bytes / 8 || n > out_f64->bytes / 8 - out64_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: out_f64 too small");
    if (out_f32 && (out32_off > out_f32->bytes / 4 || n > out_f32->bytes / 4 - out32_off)) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: out_f32 too small");
    if (!out_f64 && !out_f32 && !out_host) return sh::set_error(SH_ERR_INVALID, "sh_ew_f64: no destination");
    if (!n) return SH_OK;
    float* d32 = out_f32 ? (float*)out_f32->ptr + out32_off : nullptr;
    if (!d32 && out_host) {
        int rc = sh::ensure_scratch(n * 4);
        if (rc) return rc;
        d32 = (float*)sh::state().scratch;
    }
    hipStream_t st = sh::state().stream;
    hipLaunchKernelGGL(k_ew_f64, sh::grid1d(n, 256), dim3(256), 0, st, op,
                       need_a ? (const double*)a->ptr + a_off : nullptr, need_b ? (const double*)b->ptr + b_off : nullptr,
                       n, p0, p1, out_f64 ? (double*)out_f64->ptr + out64_off : nullptr, d32);
    SH_CHECK_LAUNCH("k_ew_f64");
    if (out_host) {
        SH_HIP(hipMemcpyAsync(out_host, d32, n * 4, hipMemcpyDeviceToHost, st));
        SH_HIP(hipStreamSynchronize(st));
    }
    return SH_OK;
}

int sh_scan_rows_f64(sh_buf* rows, size_t row0, uint32_t nrows, uint32_t n, size_t row_stride, sh_buf* carry) {
    SH_REQUIRE_INIT();
    if (!rows || !carry) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: NULL argument");
    if (!nrows || !n) return SH_OK;
    if (row_stride < n || rows->bytes / 8 < (row0 + nrows - 1) * row_stride + n) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: rows buffer too small");
    if (carry->bytes / 8 < nrows) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: carry buffer smaller than nrows doubles");
    if (nrows > 65535) return sh::set_error(SH_ERR_INVALID, "sh_scan_rows_f64: more than 65535 rows");
    const uint32_t ntiles = sh::div_up(n, SCAN_TILE);
    int rc = sh::ensure_scratch((size_t)nrows * (ntiles + 1) * 8);
    if (rc) return rc;
    double* sums = (double*)sh::state().scratch;
    double* x = (double*)rows->ptr + row0 * row_stride;
    hipStream_t st = sh::state().stream;
    hipLaunchKernelGGL(k_scan_rows_tile_sums, dim3(ntiles, nrows), dim3(256), 0, st, (const double*)x, row_stride, n, ntiles, sums);
    SH_CHECK_LAUNCH("k_scan_rows_tile_sums");
    hipLaunchKernelGGL(k_scan_rows_sums, dim3(nrows), dim3(256), 0, st, sums, ntiles, (double*)carry->ptr);
    SH_CHECK_LAUNCH("k_scan_rows_sums");
    hipLaunchKernelGGL(k_scan_rows_apply, dim3(ntiles, nrows), dim3(256), 0, st, (const double*)x, row_stride, n, ntiles, (const double*)sums, x);
    SH_CHECK_LAUNCH("k_scan_rows_apply");
    return SH_OK;
}

int sh_scan_f64(const sh_buf* x, uint32_t n, double carry_in, sh_buf* out, double* carry_out) {
    SH_REQUIRE_INIT();
    if (!x || !out) return sh::set_error(SH_ERR_INVALID, "sh_scan_f64: NULL argument");
    if (x->bytes < (size_t)n * 8 || out->bytes < (size_t)n * 8) return sh::set_error(SH_ERR_INVALID, "sh_scan_f64: buffer too small");
    if (n == 0) {
        if (carry_out) *carry_out = carry_in;
        return SH_OK;
    }
    uint32_t ntiles = sh::div_up(n, SCAN_TILE);
    int rc = sh::ensure_scratch((size_t)(ntiles + 1) * 8);
    if (rc) return rc;
    double* sums = (double*)sh::state().scratch;
    hipStream_t st = sh::state().stream;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(256), 0, st, (const double*)x->ptr, n, sums);
    SH_CHECK_LAUNCH("k_scan_tile_sums");
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, sums, ntiles, carry_in);
    SH_CHECK_LAUNCH("k_scan_sums");
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(256), 0, st, (const double*)x->ptr, n, (const double*)sums, (double*)out->ptr);
    SH_CHECK_LAUNCH("k_scan_apply");
    if (carry_out) {
        SH_HIP(hipMemcpyAsync(carry_out, sums + ntiles, 8, hipMemcpyDeviceToHost, st));
        SH_HIP(hipStreamSynchronize(st));
    }
    return SH_OK;
}

}  // extern "C"
