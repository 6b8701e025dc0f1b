// Small host-side utilities of the stand-alone driver: RAII device buffers and the vector-kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace exa_host {

constexpr int DOT_BLOCKS = 1024;

inline void hip_check(hipError_t e, const char* what) {
   if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define EXA_HC(call) ::exa_host::hip_check((call), #call)

template <typename T>
struct DevBuf {
   T* p = nullptr; size_t n = 0;
   DevBuf() = default;
   explicit DevBuf(size_t n_) { alloc(n_); }
   DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
   DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
   DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
   ~DevBuf() { release(); }
   void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
   void alloc(size_t n_) { release(); n = n_; if (n) EXA_HC(hipMalloc(&p, sizeof(T) * n)); }
   void zero(hipStream_t s = nullptr) { if (n) EXA_HC(hipMemsetAsync(p, 0, sizeof(T) * n, s)); }
   void upload(const T* h, size_t cnt, hipStream_t s = nullptr) { EXA_HC(hipMemcpyAsync(p, h, sizeof(T) * cnt, hipMemcpyHostToDevice, s)); EXA_HC(hipStreamSynchronize(s)); }
   void upload(const std::vector<T>& h, hipStream_t s = nullptr) { if (n < h.size()) alloc(h.size()); if (!h.empty()) upload(h.data(), h.size(), s); }
   void download(T* h, size_t cnt, hipStream_t s = nullptr) const { EXA_HC(hipMemcpyAsync(h, p, sizeof(T) * cnt, hipMemcpyDeviceToHost, s)); EXA_HC(hipStreamSynchronize(s)); }
   std::vector<T> to_host(hipStream_t s = nullptr) const { std::vector<T> h(n); if (n) download(h.data(), n, s); return h; }
   void copy_from(const DevBuf<T>& o, hipStream_t s = nullptr) { EXA_HC(hipMemcpyAsync(p, o.p, sizeof(T) * o.n, hipMemcpyDeviceToDevice, s)); }
   void swap(DevBuf<T>& o) { std::swap(p, o.p); std::swap(n, o.n); }
};

void vk_poison_if_failed(const int* fail_count, double* sum, double* count_out, hipStream_t s);
void vk_update_coords(int64_t n, const double* xb, const double* v, double dt, double* xe, hipStream_t s);
void vk_mask_zero(int64_t n, const uint8_t* m, double* y, hipStream_t s);
void vk_mask_set(int64_t n, const uint8_t* m, const double* val, double* y, hipStream_t s);
void vk_mask_one(int64_t n, const uint8_t* m, double* y, hipStream_t s);
void vk_axpby(int64_t n, double a, const double* x, double b, double* y, hipStream_t s);
void vk_jacobi_setup(int64_t n, const uint8_t* m, const double* diag, int identity, double* dinv, hipStream_t s);
void vk_pointwise(int64_t n, const double* a, const double* b, double* y, hipStream_t s);
void vk_fill_if(int64_t n, const double* flag, double val, double* y, hipStream_t s);
void vk_dot(int64_t n, int64_t nn, const double* w, const double* a, const double* b, const double* flag, double* partial, double* out, hipStream_t s);
void vk_cg_init(double* S, double rel, double abs_, hipStream_t s);
void vk_cg_den(double* S, hipStream_t s);
void vk_cg_beta(double* S, int max_iter, hipStream_t s);
void vk_cg_step1(int64_t n, int64_t nn, double* S, const double* w, const double* dinv, const double* d, double* x, double* r, double* z, double* partial, bool ident,
                 bool fuse_beta, int max_iter, hipStream_t s, const double* partialD = nullptr);   // partialD: consumer-side reduction of the denominator (vec_kernels.hip)
void vk_cg_step2(int64_t n, const double* S, const double* z, double* d, hipStream_t s);
// single-reduction PCG (more than one rank): see vec_kernels.hip
void vk_cg2_init(double* S, double rel, double abs_, hipStream_t s);
void vk_cg2_scalars(double* S, int max_iter, hipStream_t s);
void vk_cg2_update(int64_t n, const double* S, const double* dinv, double* x, double* r, double* u, double* p, double* sv, double* q, bool ident, hipStream_t s);
void vk_cg2_dots(int64_t n, int64_t nn, const double* w, const uint8_t* m, const double* r, const double* u, double* sv, const double* flag, double* partial, double* out2,
                 bool ident, hipStream_t s);
void vk_cg_step2z(int64_t n, double* S, double* z, const double* r, double* d, bool ident, hipStream_t s, const double* partialN = nullptr, int max_iter = 0);   // ... and z = 0 for the accumulating operator action; partialN: consumer-side reduction of (r, z)
void vk_dot_partial(int64_t n, int64_t nn, const double* w, const double* a, const double* b, const double* flag, double* partial, hipStream_t s);   // partial sums only (their consumer reduces them)
void vk_mask_dot(int64_t n, int64_t nn, const double* w, const uint8_t* m, const double* a, double* b, const double* flag, double* partial, double* out, hipStream_t s,
                 double* fuse_den_S = nullptr);
void vk_min3(int64_t nn, const double* x, double* partial /*>= DOT_BLOCKS*/, double* out3, hipStream_t s);
void vk_vgrad_velocity(int64_t nn, const uint8_t* m, const double* x, const double* org3_dev, const double* L9_host, double* v, hipStream_t s);
void vk_qf_eb64_to_aos(int W, int Q, int64_t E, const double* src_eb64, double* dst_aos, hipStream_t s);   // (W, Q, E) <- [block][q][W][lane]
void vk_max_abs_diff(int64_t n, const double* a, const double* b, double* out3_dev /* max |a-b|, max |a|, (u64) differing entries of the skipped component */, hipStream_t s, int W = 0, int skip = -1);
void vk_pack(int64_t n, const int32_t* idx, const double* y, double* buf, hipStream_t s);
void vk_unpack_add(int64_t n, const int32_t* idx, const double* buf, double* y, hipStream_t s);

}  // namespace exa_host
