// Microbenchmark for DESIGN section 8 item 7: the 16-bit mono resample (audioop.ratecv's arithmetic, reduced rates below 65536) with
// the frames dealt to the lanes (csrc/pcm.hip k_resample_mono16), its input (A) staged through LDS behind a barrier, as shipped, or
// (B) read straight from global memory -- each lane its two dwords, neighbouring lanes the same or neighbouring ones, so the L1 sees
// two lines per wave instruction -- with no LDS, no barrier and no workgroup structure at all.  Same outputs (compared on the
// device), 450 M input frames, 44.1 -> 48 kHz, 96 -> 44.1 kHz, 48 -> 44.1 kHz.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { uint32_t inr, outr, step_q, step_r; double inv_outr; };

__device__ __forceinline__ void place(const Args& A, uint64_t m, uint64_t& q, uint32_t& r) {   // frame m sits at input position q + r/outr
    const uint64_t M = m * (uint64_t)A.inr;
    uint64_t qq = (uint64_t)floor((double)M * A.inv_outr);
    int64_t rr = (int64_t)(M - qq * (uint64_t)A.outr);
    if (rr < 0) { qq -= 1; rr += A.outr; } else if (rr >= (int64_t)A.outr) { qq += 1; rr -= A.outr; }
    q = qq; r = (uint32_t)rr;
}

__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (uint32_t)(z ^ (z >> 31));
    }
}

__device__ __forceinline__ uint32_t interp(uint32_t lo, uint32_t hi, uint32_t qel, uint32_t Rb, const Args& A, double half_inv) {
    const uint32_t pair = __builtin_amdgcn_alignbit(hi, lo, qel << 4) ^ 0x80008000u;
    const uint32_t ua = pair & 0xffffu, ub = pair >> 16;
    const uint32_t u = (uint32_t)__mul24((int)ub - (int)ua, (int)Rb) + __umul24(ub, A.outr);
    return (uint32_t)fma((double)u, A.inv_outr, half_inv);
}
__device__ __forceinline__ void advance(uint32_t& Rb, uint32_t& qel, uint32_t sr, uint32_t sq, uint32_t outr) {
    uint32_t Rn;
    const uint32_t c = __builtin_uadd_overflow(Rb, sr, &Rn) ? 1u : 0u;
    qel += sq + c;
    Rb = max(Rn, Rn - outr);
}

// (A) as shipped (WAVES = 4): the workgroup's span through LDS; WAVES = 1 .. 16: the same with 1024 WAVES frames per workgroup
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_staged(const short* __restrict__ in, short* __restrict__ out, Args A, uint32_t span_vecs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef short ld_t __attribute__((ext_vector_type(8)));
    const uint64_t m_first = (uint64_t)blockIdx.x * (1024 * WAVES);
    uint64_t q0; uint32_t r0;
    place(A, m_first, q0, r0);
    const uint64_t lo_elem = q0 & ~7ull;
    for (uint32_t v = threadIdx.x; v < span_vecs; v += 64 * WAVES)
        reinterpret_cast<ld_t*>(smem)[v] = __builtin_nontemporal_load(reinterpret_cast<const ld_t*>(in + lo_elem + (uint64_t)v * 8));
    __syncthreads();
    const uint32_t f0 = (threadIdx.x >> 6) * 1024u + 2u * (threadIdx.x & 63u);
    const double half_inv = 0.5 * A.inv_outr;
    const uint32_t tot = r0 + __umul24(f0, A.inr);
    const uint32_t dq = (uint32_t)fma((double)tot, A.inv_outr, half_inv);
    uint32_t R = tot - dq * A.outr - A.outr, qe = (uint32_t)(q0 - lo_elem) + dq;
    const uint32_t t127 = 127u * A.inr, q127 = (uint32_t)fma((double)t127, A.inv_outr, half_inv), r127 = t127 - q127 * A.outr;
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out + m_first + f0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t* l = reinterpret_cast<const uint32_t*>(smem) + (qe >> 1);
        const uint32_t a = interp(l[0], l[1], qe, R, A, half_inv);
        advance(R, qe, A.step_r, A.step_q, A.outr);
        l = reinterpret_cast<const uint32_t*>(smem) + (qe >> 1);
        const uint32_t b = interp(l[0], l[1], qe, R, A, half_inv);
        if (s < 7) advance(R, qe, r127, q127, A.outr);
        __builtin_nontemporal_store((a | (b << 16)) ^ 0x80008000u, o32 + 64 * s);
    }
}

// (B) no staging: a wave = 1024 consecutive output frames, every lane loads its own two dwords per frame
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_direct(const short* __restrict__ in, short* __restrict__ out, Args A) {
    const uint64_t wave = (uint64_t)blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t m_first = wave * 1024;
    uint64_t q0; uint32_t r0;
    place(A, m_first, q0, r0);                                    // (uniform per wave)
    const double half_inv = 0.5 * A.inv_outr;
    const uint32_t tot = r0 + __umul24(2u * lane, A.inr);
    const uint32_t dq = (uint32_t)fma((double)tot, A.inv_outr, half_inv);
    uint32_t R = tot - dq * A.outr - A.outr, qe = (uint32_t)(q0 & 1ull) + dq;       // element index relative to the dword that holds q0
    const uint32_t* base = reinterpret_cast<const uint32_t*>(in) + (q0 >> 1);
    const uint32_t t127 = 127u * A.inr, q127 = (uint32_t)fma((double)t127, A.inv_outr, half_inv), r127 = t127 - q127 * A.outr;
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out + m_first + 2u * lane);
    // all sixteen dword pairs first (their addresses need only the position walk), then the arithmetic: the loads are in flight together
    uint32_t lo[16], hi[16], qs[16], Rs[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        qs[2 * s] = qe; Rs[2 * s] = R;
        advance(R, qe, A.step_r, A.step_q, A.outr);
        qs[2 * s + 1] = qe; Rs[2 * s + 1] = R;
        if (s < 7) advance(R, qe, r127, q127, A.outr);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t* g = base + (qs[k] >> 1);
        lo[k] = __builtin_nontemporal_load(g);
        hi[k] = __builtin_nontemporal_load(g + 1);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t a = interp(lo[2 * s], hi[2 * s], qs[2 * s], Rs[2 * s], A, half_inv);
        const uint32_t b = interp(lo[2 * s + 1], hi[2 * s + 1], qs[2 * s + 1], Rs[2 * s + 1], A, half_inv);
        __builtin_nontemporal_store((a | (b << 16)) ^ 0x80008000u, o32 + 64 * s);
    }
}

__global__ void k_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* bad) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

static uint32_t gcd(uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; }

int main() {
    const size_t in_frames = 450000000;
    short *in, *oa, *ob;
    unsigned long long* bad;
    CK(hipMalloc(&in, in_frames * 2 + 4096));
    CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)in, in_frames / 2 + 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t rates[3][2] = {{44100, 48000}, {96000, 44100}, {48000, 44100}};
    for (auto& rt : rates) {
        const uint32_t g = gcd(rt[0], rt[1]);
        Args A;
        A.inr = rt[0] / g; A.outr = rt[1] / g; A.step_q = A.inr / A.outr; A.step_r = A.inr % A.outr; A.inv_outr = 1.0 / (double)A.outr;
        size_t nout = (size_t)(((unsigned __int128)(in_frames - 1) * A.outr) / A.inr) + 1;
        nout &= ~(size_t)4095;                                     // whole workgroups only: the tails are not what is measured
        CK(hipMalloc(&oa, nout * 2)); CK(hipMalloc(&ob, nout * 2));
        float msa = 0, msb = 0, msb2 = 0, msw[5] = {0, 0, 0, 0, 0};
        auto spanv = [&](uint32_t frames) { const uint64_t sf = ((uint64_t)frames * A.inr + A.outr - 1) / A.outr + 3; return (uint32_t)((sf + 8 + 7) / 8 + 1); };
        nout &= ~(size_t)16383;
        for (int rep = 0; rep < 12; ++rep) {
            float t;
#define TIME(DST, ...) CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); if (rep >= 4 && (DST == 0 || t < DST)) DST = t;
            TIME(msa, hipLaunchKernelGGL(k_staged<4>, dim3((uint32_t)(nout / 4096)), dim3(256), spanv(4096) * 16, 0, in, oa, A, spanv(4096)))
            TIME(msb, hipLaunchKernelGGL(k_direct<4>, dim3((uint32_t)(nout / 4096)), dim3(256), 0, 0, in, ob, A))
            TIME(msb2, hipLaunchKernelGGL(k_direct<1>, dim3((uint32_t)(nout / 1024)), dim3(64), 0, 0, in, ob, A))
            TIME(msw[0], hipLaunchKernelGGL(k_staged<1>, dim3((uint32_t)(nout / 1024)), dim3(64), spanv(1024) * 16, 0, in, ob, A, spanv(1024)))
            TIME(msw[1], hipLaunchKernelGGL(k_staged<2>, dim3((uint32_t)(nout / 2048)), dim3(128), spanv(2048) * 16, 0, in, ob, A, spanv(2048)))
            TIME(msw[2], hipLaunchKernelGGL(k_staged<8>, dim3((uint32_t)(nout / 8192)), dim3(512), spanv(8192) * 16, 0, in, ob, A, spanv(8192)))
            TIME(msw[3], hipLaunchKernelGGL(k_staged<16>, dim3((uint32_t)(nout / 16384)), dim3(1024), spanv(16384) * 16, 0, in, ob, A, spanv(16384)))
        }
        printf("   staged with 1 / 2 / 8 / 16 waves per workgroup: %.3f / %.3f / %.3f / %.3f ms\n", msw[0], msw[1], msw[2], msw[3]);
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, 0, (const uint32_t*)oa, (const uint32_t*)ob, nout / 2, bad);
        unsigned long long hb = 0;
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        const double bytes = (double)(in_frames + nout) * 2;
        printf("%u -> %u: staged %.3f ms (%.3f of 8 TB/s)  direct, 4 waves per workgroup %.3f ms (%.3f)  direct, 1 wave %.3f ms (%.3f)  differing dwords %llu\n",
               rt[0], rt[1], msa, bytes / (msa * 1e-3) / 8e12, msb, bytes / (msb * 1e-3) / 8e12, msb2, bytes / (msb2 * 1e-3) / 8e12, hb);
        CK(hipFree(oa)); CK(hipFree(ob));
    }
    return 0;
}
