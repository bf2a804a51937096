// The step before the hot path: center-surround
// whitening of whole images in the frequency domain and extraction of training
// patches, on the device, so that patches never cross PCIe.
//
// Restates
//   utils/image_processing.py:63-92    filter_fd
//   utils/image_processing.py:173-231  get_low_pass_filter ('exponential')
//   utils/image_processing.py:234-264  get_whitening_ramp_filter
//   utils/image_processing.py:267-308  whiten_center_surround
//   utils/dataset_generation.py:184-222  the 'patch' operation
//
// The reference filters in float64 (numpy FFT) and casts the result to
// float32; so does this file (hipFFT D2Z / Z2D; the transfer function is real
// and even, so the half-spectrum transform is exact).  hipFFT is opened with
// dlopen at first use: the rest of the library has no dependency on it.
#include "common.h"

#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <hipfft/hipfft.h>

#include <map>
#include <mutex>
#include <tuple>

namespace vtc {

static unsigned flat_grid(int64_t total) {
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 65536) blocks = 65536;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

struct FftApi {
  hipfftResult (*plan_many)(hipfftHandle*, int, int*, int*, int, int, int*,
                            int, int, hipfftType, int);
  hipfftResult (*exec_d2z)(hipfftHandle, hipfftDoubleReal*,
                           hipfftDoubleComplex*);
  hipfftResult (*exec_z2d)(hipfftHandle, hipfftDoubleComplex*,
                           hipfftDoubleReal*);
  hipfftResult (*set_stream)(hipfftHandle, hipStream_t);
  bool ok;
};

static const FftApi& fft_api() {
  static FftApi api = [] {
    FftApi a{};
    // A process that imported PyTorch already holds PyTorch's own hipFFT
    // (and HIP runtime); a second copy from /opt/rocm would bring a second
    // runtime with it.  Take the loaded one if there is one.
    void* lib = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("libhipfft.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) {
      if (FILE* maps = fopen("/proc/self/maps", "r")) {
        char line[1024];
        while (!lib && fgets(line, sizeof line, maps)) {
          char* path = strchr(line, '/');
          if (!path || !strstr(path, "libhipfft.so")) continue;
          path[strcspn(path, "\n")] = 0;
          lib = dlopen(path, RTLD_NOW | RTLD_NOLOAD);
        }
        fclose(maps);
      }
    }
    if (!lib) lib = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("libhipfft.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return a;
    a.plan_many = reinterpret_cast<decltype(a.plan_many)>(
        dlsym(lib, "hipfftPlanMany"));
    a.exec_d2z = reinterpret_cast<decltype(a.exec_d2z)>(
        dlsym(lib, "hipfftExecD2Z"));
    a.exec_z2d = reinterpret_cast<decltype(a.exec_z2d)>(
        dlsym(lib, "hipfftExecZ2D"));
    a.set_stream = reinterpret_cast<decltype(a.set_stream)>(
        dlsym(lib, "hipfftSetStream"));
    a.ok = a.plan_many && a.exec_d2z && a.exec_z2d && a.set_stream;
    return a;
  }();
  return api;
}

struct FftPlans {
  hipfftHandle forward, inverse;
};

// plans are cached per (device, h, w, batch)
static int get_plans(int h, int w, int batch, FftPlans* out) {
  static std::mutex lock;
  static std::map<std::tuple<int, int, int, int>, FftPlans> cache;
  const FftApi& api = fft_api();
  if (!api.ok) {
    set_error("whitening: libhipfft.so could not be opened");
    return VTC_ERR_UNSUPPORTED;
  }
  int device = 0;
  VTC_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> guard(lock);
  const auto key = std::make_tuple(device, h, w, batch);
  auto it = cache.find(key);
  if (it == cache.end()) {
    FftPlans p;
    int n[2] = {h, w};
    if (api.plan_many(&p.forward, 2, n, nullptr, 1, 0, nullptr, 1, 0,
                      HIPFFT_D2Z, batch) != HIPFFT_SUCCESS ||
        api.plan_many(&p.inverse, 2, n, nullptr, 1, 0, nullptr, 1, 0,
                      HIPFFT_Z2D, batch) != HIPFFT_SUCCESS) {
      set_error("whitening: hipfftPlanMany failed for %dx%d x %d", h, w,
                batch);
      return VTC_ERR_HIP;
    }
    it = cache.emplace(key, p).first;
  }
  *out = it->second;
  return VTC_OK;
}

// (count, h, w, c) float32 channel-last -> (count*c, h, w) float64 planes
__global__ void planes_from_images_kernel(const float* __restrict__ in,
                                          double* __restrict__ out,
                                          int64_t count, int h, int w, int c) {
  const int64_t total = count * c * h * w;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int ch = (int)((i / ((int64_t)w * h)) % c);
    const int64_t img = i / ((int64_t)w * h * c);
    out[i] = (double)in[((img * h + y) * w + x) * c + ch];
  }
}

__global__ void images_from_planes_kernel(const double* __restrict__ in,
                                          float* __restrict__ out,
                                          int64_t count, int h, int w, int c) {
  const int64_t total = count * c * h * w;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int ch = (int)((i / ((int64_t)w * h)) % c);
    const int64_t img = i / ((int64_t)w * h * c);
    out[((img * h + y) * w + x) * c + ch] = (float)in[i];
  }
}

// numpy.fft.fftfreq(n)[k]
__device__ __forceinline__ double fft_freq(int k, int n) {
  return (double)(k < (n + 1) / 2 ? k : k - n) / (double)n;
}

// the transfer function at frequency bin (ky, kx):
//   ramp    = max(|f|, low)                     image_processing.py:255-264,298
//   lowpass = exp(-(|f| / (0.5 high))^8)        image_processing.py:219-221
__device__ __forceinline__ double whitening_gain(int ky, int kx, int h, int w,
                                                 double low, double high) {
  const double fy = fft_freq(ky, h), fx = fft_freq(kx, w);
  const double mag = sqrt(fy * fy + fx * fx);
  const double ramp = fmax(mag, low);
  const double lpf = exp(-1. * pow(mag / (0.5 * high), 8.0));
  return ramp * lpf;
}

// norm_and_threshold (image_processing.py:302-304): the maximum of the
// transfer function over the grid (it is even in both frequencies, so the
// half spectrum holds every value) as the bits of a non-negative double
__global__ void whitening_gain_max_kernel(int h, int w, double low, double high,
                                          unsigned long long* __restrict__ out) {
  const int wh = w / 2 + 1;
  const int64_t total = (int64_t)h * wh;
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    m = fmax(m, fabs(whitening_gain((int)(i / wh), (int)(i % wh), h, w, low,
                                    high)));
  m = fmax(m, __shfl_xor(m, 32, 64));
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0)
    atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// spectrum (planes, h, w/2+1) *= transfer function / (h w) -- the 1/(h w) of
// the unnormalised inverse transform.  gain_max != null: norm_and_threshold,
// i.e. the transfer function divided by its maximum, values below 1e-3
// raised to 1e-3 (image_processing.py:302-304).
__global__ void whitening_filter_kernel(hipfftDoubleComplex* __restrict__ spec,
                                        int64_t planes, int h, int w,
                                        double low, double high,
                                        const unsigned long long* gain_max) {
  const int wh = w / 2 + 1;
  const int64_t total = planes * h * wh;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const double norm = 1.0 / ((double)h * (double)w);
  const double top = gain_max ? __longlong_as_double((long long)*gain_max) : 1.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int kx = (int)(i % wh);
    const int ky = (int)((i / wh) % h);
    double gain = whitening_gain(ky, kx, h, w, low, high);
    if (gain_max) {
      gain = gain / top;
      if (fabs(gain) < 1e-3) gain = 1e-3;
    }
    const double f = gain * norm;
    spec[i].x *= f;
    spec[i].y *= f;
  }
}

// (x - lo) / (hi - lo) in float32, lo / hi from device memory
// (dataset_generation.py:169-183, 'standardize_data_range')
__global__ void standardize_range_kernel(const float* __restrict__ x,
                                         float* __restrict__ out, int64_t count,
                                         const float* __restrict__ min_max) {
  const float lo = min_max[0];
  const float span = sub_rn(min_max[1], lo);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = sub_rn(x[i], lo) / span;
}

// ---- numpy's legacy generator on the host (MT19937) ----------------------
// dataset_generation.py:205-214 draws three np.random.randint values per patch
// -- 393 216 interpreter round trips for one batch of 131 072 patches, several
// times longer than the training step they feed.  The same numbers come out
// of this loop: MT19937 (Matsumoto & Nishimura 1998; key of 624 words + a
// position, exactly numpy.random.RandomState.get_state()), and randint's
// masked rejection sampling for ranges that fit 32 bits (one 32-bit output per
// attempt; a range of one value consumes nothing).
struct Mt19937 {
  uint32_t* key;
  int pos;
  uint32_t next() {
    if (pos >= 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (key[i] & 0x80000000u) | (key[(i + 1) % 624] &
                                                      0x7fffffffu);
        key[i] = key[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      pos = 0;
    }
    uint32_t y = key[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // numpy.random.RandomState.randint(low, high) for high - low <= 2^32
  int64_t randint(int64_t low, int64_t high) {
    const uint32_t rng = (uint32_t)(high - low - 1);
    if (rng == 0) return low;
    if (rng == 0xffffffffu) return low + (int64_t)next();
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
    mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = next() & mask) > rng) {}
    return low + (int64_t)v;
  }
};

// patches[p, (dy, dx, ch)] = images[img[p], vert[p] + dy, horz[p] + dx, ch]
__global__ __launch_bounds__(256) void extract_patches_kernel(
    const float* __restrict__ images, const int32_t* __restrict__ img_index,
    const int32_t* __restrict__ vert, const int32_t* __restrict__ horz,
    float* __restrict__ patches, int64_t num, int h, int w, int c, int ph,
    int pw) {
  const int row_len = pw * c;
  const int n = ph * row_len;
  const int64_t total = num * n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int64_t p = i / n;
    const int e = (int)(i % n);
    const int dy = e / row_len, rest = e % row_len;
    const int64_t src = (((int64_t)img_index[p] * h + vert[p] + dy) * w +
                         horz[p]) * c + rest;
    patches[i] = images[src];
  }
}

}  // namespace vtc

using namespace vtc;

extern "C" size_t vtc_whiten_center_surround_workspace_bytes(int64_t count,
                                                             int32_t h,
                                                             int32_t w,
                                                             int32_t c) {
  if (count <= 0 || h <= 0 || w <= 0 || c <= 0) return 256;
  const size_t planes = (size_t)count * c;
  return align_up(planes * h * w * sizeof(double), 256) +
         align_up(planes * h * (w / 2 + 1) * sizeof(hipfftDoubleComplex), 256) +
         256;   // maximum of the transfer function (norm_and_threshold)
}

extern "C" int vtc_whiten_center_surround(const float* images, float* out,
                                          int64_t count, int32_t h, int32_t w,
                                          int32_t c, float cutoff_low,
                                          float cutoff_high,
                                          int norm_and_threshold,
                                          void* workspace,
                                          size_t workspace_bytes,
                                          void* stream) {
  VTC_REQUIRE(images && out, "vtc_whiten_center_surround: null pointer");
  VTC_REQUIRE(count > 0 && h > 0 && w > 0 && c > 0,
              "vtc_whiten_center_surround: bad sizes");
  VTC_REQUIRE(cutoff_high > 0.f && cutoff_high <= 1.f,
              "vtc_whiten_center_surround: 'high' cutoff must be in (0, 1]");
  VTC_REQUIRE(count * c <= 0x7fffffffLL,
              "vtc_whiten_center_surround: too many planes");
  if (!workspace || workspace_bytes < vtc_whiten_center_surround_workspace_bytes(
                                          count, h, w, c)) {
    set_error("vtc_whiten_center_surround: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  const int64_t planes = count * c;
  FftPlans plans;
  int rc = get_plans(h, w, (int)planes, &plans);
  if (rc != VTC_OK) return rc;
  const FftApi& api = fft_api();
  hipStream_t st = as_stream(stream);
  Carver ws(workspace);
  double* real = ws.take<double>((size_t)planes * h * w);
  hipfftDoubleComplex* spec =
      ws.take<hipfftDoubleComplex>((size_t)planes * h * (w / 2 + 1));
  unsigned long long* gain_max = nullptr;
  if (norm_and_threshold) {
    gain_max = ws.take<unsigned long long>(1);
    VTC_HIP_CHECK(hipMemsetAsync(gain_max, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(whitening_gain_max_kernel,
                       dim3(flat_grid((int64_t)h * (w / 2 + 1))), dim3(256), 0,
                       st, h, w, (double)cutoff_low, (double)cutoff_high,
                       gain_max);
    VTC_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(planes_from_images_kernel,
                     dim3(flat_grid(planes * h * w)), dim3(256), 0, st, images,
                     real, count, h, w, c);
  VTC_LAUNCH_CHECK();
  if (api.set_stream(plans.forward, st) != HIPFFT_SUCCESS ||
      api.set_stream(plans.inverse, st) != HIPFFT_SUCCESS ||
      api.exec_d2z(plans.forward, real, spec) != HIPFFT_SUCCESS) {
    set_error("vtc_whiten_center_surround: forward transform failed");
    return VTC_ERR_HIP;
  }
  hipLaunchKernelGGL(whitening_filter_kernel,
                     dim3(flat_grid(planes * h * (w / 2 + 1))), dim3(256), 0,
                     st, spec, planes, h, w, (double)cutoff_low,
                     (double)cutoff_high, gain_max);
  VTC_LAUNCH_CHECK();
  if (api.exec_z2d(plans.inverse, spec, real) != HIPFFT_SUCCESS) {
    set_error("vtc_whiten_center_surround: inverse transform failed");
    return VTC_ERR_HIP;
  }
  hipLaunchKernelGGL(images_from_planes_kernel,
                     dim3(flat_grid(planes * h * w)), dim3(256), 0, st, real,
                     out, count, h, w, c);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" int vtc_extract_patches(const float* images,
                                   const int32_t* img_index,
                                   const int32_t* vert, const int32_t* horz,
                                   float* patches, int64_t num, int32_t h,
                                   int32_t w, int32_t c, int32_t ph,
                                   int32_t pw, void* stream) {
  VTC_REQUIRE((images && img_index && vert && horz && patches) || num == 0,
              "vtc_extract_patches: null pointer");
  VTC_REQUIRE(num >= 0 && h > 0 && w > 0 && c > 0 && ph > 0 && pw > 0 &&
                  ph <= h && pw <= w, "vtc_extract_patches: bad sizes");
  if (num == 0) return VTC_OK;
  hipLaunchKernelGGL(extract_patches_kernel,
                     dim3(flat_grid(num * ph * pw * c)), dim3(256), 0,
                     as_stream(stream), images, img_index, vert, horz, patches,
                     num, h, w, c, ph, pw);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// (x - min) / (max - min) over the whole array; min_max (device, 2 floats)
// receives the two extremes (the reference asserts max > min: the caller
// reads them back if it wants to).  workspace: vtc_window_minmax's.
extern "C" int vtc_window_minmax(const float* x, int64_t outer, int64_t rows,
                                 int64_t cols, int64_t outer_pitch,
                                 int64_t row_pitch, float* out_min_max,
                                 void* workspace, size_t workspace_bytes,
                                 void* stream);

extern "C" int vtc_standardize_data_range(const float* images, float* out,
                                          int64_t count, float* min_max,
                                          void* workspace,
                                          size_t workspace_bytes,
                                          void* stream) {
  VTC_REQUIRE(images && out && min_max,
              "vtc_standardize_data_range: null pointer");
  VTC_REQUIRE(count > 0, "vtc_standardize_data_range: empty array");
  int rc = vtc_window_minmax(images, 1, 1, count, 0, 0, min_max, workspace,
                             workspace_bytes, stream);
  if (rc != VTC_OK) return rc;
  hipLaunchKernelGGL(standardize_range_kernel, dim3(flat_grid(count)),
                     dim3(256), 0, as_stream(stream), images, out, count,
                     min_max);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// Host-side: the patch positions of dataset_generation.py:205-214 from numpy's
// legacy generator state.  key (624 words) and *pos are
// RandomState.get_state()[1:3], updated in place so that the caller can hand
// them back with set_state.  Per patch: image index in [0, num_images),
// vertical position in [edge_buffer, max_vert[image]), horizontal position in
// [edge_buffer, max_horz[image]) -- in that order, as the reference draws
// them.  img_index / vert / horz: num_samples int32 each (HOST memory).
extern "C" int vtc_draw_patch_positions(uint32_t* key, int32_t* pos,
                                        int64_t num_samples,
                                        int32_t num_images,
                                        int32_t edge_buffer,
                                        const int32_t* max_vert,
                                        const int32_t* max_horz,
                                        int32_t* img_index, int32_t* vert,
                                        int32_t* horz) {
  VTC_REQUIRE(key && pos && max_vert && max_horz && img_index && vert && horz,
              "vtc_draw_patch_positions: null pointer");
  VTC_REQUIRE(num_samples >= 0 && num_images > 0 && *pos >= 0 && *pos <= 624,
              "vtc_draw_patch_positions: bad sizes or generator position");
  for (int i = 0; i < num_images; ++i)
    VTC_REQUIRE(max_vert[i] > edge_buffer && max_horz[i] > edge_buffer,
                "vtc_draw_patch_positions: image %d is smaller than a patch "
                "plus its edge buffer", i);
  Mt19937 mt{key, *pos};
  for (int64_t p = 0; p < num_samples; ++p) {
    const int32_t img = (int32_t)mt.randint(0, num_images);
    img_index[p] = img;
    vert[p] = (int32_t)mt.randint(edge_buffer, max_vert[img]);
    horz[p] = (int32_t)mt.randint(edge_buffer, max_horz[img]);
  }
  *pos = mt.pos;
  return VTC_OK;
}
