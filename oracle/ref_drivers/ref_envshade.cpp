// Host driver for the reference's OptiX raygen program (TEST INFRASTRUCTURE -- checker only).
//
// Compiles /root/reference/render/optixutils/c_src/envsampling/kernel.cu (with its bsdf.h, math_utils.h, common.h,
// accessor.h, params.h) unchanged, through oracle/ref_stub, and drives `__raygen__rg` with a host loop over the launch grid
// exactly as `optixLaunch(..., ro.size(2), ro.size(1), ro.size(0))` does (c_src/torch_bindings.cpp:183-184, :260-261).
// Parameter set-up mirrors env_shade_fwd / env_shade_bwd (torch_bindings.cpp:123-189, :191-266): zero-initialised outputs,
// `backward` 0 / 1, gradient tensors sized like the inputs.  C ABI: plain pointers and sizes, contiguous fp32 / int32.
#include <optix.h>
#include <envsampling/kernel.cu>

#include <initializer_list>
#include <omp.h>

namespace {
template <class T, size_t N>
struct AccMaker : PackedTensorAccessor32<T, N> {       // the host-side constructors are compiled out under __CUDACC__
    AccMaker(T* d, std::initializer_list<int> sizes) {
        this->data_ = d;
        int i = 0;
        for (int s : sizes) this->sizes_[i++] = s;
        int st = 1;
        for (int k = (int)N - 1; k >= 0; --k) { this->strides_[k] = st; st *= this->sizes_[k]; }
    }
};
template <class T, size_t N>
PackedTensorAccessor32<T, N> acc(const T* d, std::initializer_list<int> sizes) {
    AccMaker<T, N> m(const_cast<T*>(d), sizes);
    return m;
}

std::vector<float> g_tri;

void set_scene(const float* verts, const int* tris, long long T) {
    g_tri.resize((size_t)T * 9);
    for (long long t = 0; t < T; ++t) {
        const float* a = verts + 3 * (long long)tris[3 * t];
        const float* b = verts + 3 * (long long)tris[3 * t + 1];
        const float* c = verts + 3 * (long long)tris[3 * t + 2];
        float* r = g_tri.data() + 9 * t;
        for (int k = 0; k < 3; ++k) { r[k] = a[k]; r[3 + k] = b[k] - a[k]; r[6 + k] = c[k] - a[k]; }
    }
    optixhost::g_scene.v0e1e2 = g_tri.data();
    optixhost::g_scene.T = T;
    ah_grid_free(&optixhost::g_scene.grid);
    if (optixhost::g_scene.use_grid) ah_grid_build(&optixhost::g_scene.grid, g_tri.data(), T);   // not built (degenerate scene): brute force
}

struct Args {
    const float *mask, *ro, *gb_pos, *gb_normal, *view_pos, *gb_kd, *gb_ks, *light, *pdf, *rows, *cols;
    const int* perms;
    int B, H, W, vB, vH, vW, Hl, Wl, n_perms, BSDF, n_samples_x;
    unsigned int rnd_seed;
    float shadow_scale;
};

void fill_common(const Args& a) {
    const int S = a.n_samples_x * a.n_samples_x;
    params.handle = 0;
    params.mask = acc<float, 3>(a.mask, {a.B, a.H, a.W});
    params.ro = acc<float, 4>(a.ro, {a.B, a.H, a.W, 3});
    params.gb_pos = acc<float, 4>(a.gb_pos, {a.B, a.H, a.W, 3});
    params.gb_normal = acc<float, 4>(a.gb_normal, {a.B, a.H, a.W, 3});
    params.gb_view_pos = acc<float, 4>(a.view_pos, {a.vB, a.vH, a.vW, 3});
    params.gb_kd = acc<float, 4>(a.gb_kd, {a.B, a.H, a.W, 3});
    params.gb_ks = acc<float, 4>(a.gb_ks, {a.B, a.H, a.W, 3});
    params.light = acc<float, 3>(a.light, {a.Hl, a.Wl, 3});
    params.pdf = acc<float, 2>(a.pdf, {a.Hl, a.Wl});
    params.rows = acc<float, 1>(a.rows, {a.Hl});
    params.cols = acc<float, 2>(a.cols, {a.Hl, a.Wl});
    params.perms = acc<int, 2>(a.perms, {a.n_perms, S});
    params.BSDF = (unsigned)a.BSDF;
    params.n_samples_x = (unsigned)a.n_samples_x;
    params.rnd_seed = a.rnd_seed;
    params.shadow_scale = a.shadow_scale;
}

void run_launch(int W, int H, int B) {
    optixhost::g_dim = uint3{(unsigned)W, (unsigned)H, (unsigned)B};
    // pixel order z, y, x: with one host thread (OMP_NUM_THREADS=1) the atomicAdd order into light_grad is fixed -- golden
    // vectors are minted that way
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int z = 0; z < B; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                optixhost::g_idx = uint3{(unsigned)x, (unsigned)y, (unsigned)z};
                __raygen__rg();
            }
}
}  // namespace

extern "C" {

// host threads of the pixel loop (1 = launch order z, y, x: deterministic atomicAdd order into light_grad)
void ref_set_threads(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); }

// any-hit evaluation of the launches that follow: 0 = every triangle (the definition), 1 = grid-filtered candidates (default)
void ref_set_anyhit_mode(int use_grid) { optixhost::g_scene.use_grid = use_grid ? 1 : 0; }

// env_shade_fwd (torch_bindings.cpp:123-189).  diff / spec [B,H,W,3] are zero-filled here like torch::zeros there.
int ref_env_shade_fwd(const float* mask, const float* ro, const float* gb_pos, const float* gb_normal, const float* view_pos,
                      const float* gb_kd, const float* gb_ks, const float* light, const float* pdf, const float* rows,
                      const float* cols, const int* perms, int B, int H, int W, int vB, int vH, int vW, int Hl, int Wl, int n_perms,
                      int BSDF, int n_samples_x, unsigned int rnd_seed, float shadow_scale, const float* verts, const int* tris,
                      long long T, float* diff, float* spec) {
    Args a{mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
           B, H, W, vB, vH, vW, Hl, Wl, n_perms, BSDF, n_samples_x, rnd_seed, shadow_scale};
    set_scene(verts, tris, T);
    fill_common(a);
    memset(diff, 0, sizeof(float) * (size_t)B * H * W * 3);
    memset(spec, 0, sizeof(float) * (size_t)B * H * W * 3);
    params.diff = acc<float, 4>(diff, {B, H, W, 3});
    params.spec = acc<float, 4>(spec, {B, H, W, 3});
    params.backward = 0;
    run_launch(W, H, B);
    return 0;
}

// env_shade_bwd (torch_bindings.cpp:191-266): five zero-initialised gradient tensors, the same launch with backward = 1.
int ref_env_shade_bwd(const float* mask, const float* ro, const float* gb_pos, const float* gb_normal, const float* view_pos,
                      const float* gb_kd, const float* gb_ks, const float* light, const float* pdf, const float* rows,
                      const float* cols, const int* perms, int B, int H, int W, int vB, int vH, int vW, int Hl, int Wl, int n_perms,
                      int BSDF, int n_samples_x, unsigned int rnd_seed, float shadow_scale, const float* verts, const int* tris,
                      long long T, const float* diff_grad, const float* spec_grad, float* gb_pos_grad, float* gb_normal_grad,
                      float* gb_kd_grad, float* gb_ks_grad, float* light_grad) {
    Args a{mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
           B, H, W, vB, vH, vW, Hl, Wl, n_perms, BSDF, n_samples_x, rnd_seed, shadow_scale};
    set_scene(verts, tris, T);
    fill_common(a);
    const size_t img = sizeof(float) * (size_t)B * H * W * 3;
    memset(gb_pos_grad, 0, img);
    memset(gb_normal_grad, 0, img);
    memset(gb_kd_grad, 0, img);
    memset(gb_ks_grad, 0, img);
    memset(light_grad, 0, sizeof(float) * (size_t)Hl * Wl * 3);
    params.diff_grad = acc<float, 4>(diff_grad, {B, H, W, 3});
    params.spec_grad = acc<float, 4>(spec_grad, {B, H, W, 3});
    params.gb_pos_grad = acc<float, 4>(gb_pos_grad, {B, H, W, 3});
    params.gb_normal_grad = acc<float, 4>(gb_normal_grad, {B, H, W, 3});
    params.gb_kd_grad = acc<float, 4>(gb_kd_grad, {B, H, W, 3});
    params.gb_ks_grad = acc<float, 4>(gb_ks_grad, {B, H, W, 3});
    params.light_grad = acc<float, 3>(light_grad, {Hl, Wl, 3});
    params.backward = 1;
    run_launch(W, H, B);
    return 0;
}

// Per-sample record of one pixel, for explaining a pixel that differs: calls the reference's own sampling functions
// (rand_pcg / lightSample / bsdf_sample / bsdf_pdf / lightPDF, kernel.cu:30-45, :184-193, :334-400) in the raygen order
// (kernel.cu:488-529) and records per sample (light and BSDF samples interleaved): direction xyz, pdf_light, pdf_bsdf,
// visibility.  out: [2*n*n, 6].  Call after a fwd / bwd launch has set `params` and the scene.
int ref_env_shade_trace_pixel(int x, int y, int z, float* out) {
    const uint3 idx = make_uint3((unsigned)x, (unsigned)y, (unsigned)z);
    const uint3 dim = optixhost::g_dim;
    float3 ray_origin = fetch3(params.ro, idx.z, idx.y, idx.x);
    float3 gb_pos = fetch3(params.gb_pos, idx.z, idx.y, idx.x);
    float3 gb_normal = fetch3(params.gb_normal, idx.z, idx.y, idx.x);
    float3 gb_view_pos = fetch3(params.gb_view_pos, idx.z, idx.y, idx.x);
    float3 gb_kd = fetch3(params.gb_kd, idx.z, idx.y, idx.x);
    float3 gb_ks = fetch3(params.gb_ks, idx.z, idx.y, idx.x);
    float strata_frac = 1.0f / params.n_samples_x;
    float alpha = gb_ks.y * gb_ks.y;
    float3 wo = safe_normalize(gb_view_pos - gb_pos);
    float metallic = gb_ks.z;
    float3 specColor = make_float3(0.04f, 0.04f, 0.04f) * (1.0f - metallic) + gb_kd * metallic;
    float diffuseWeight = (1.f - metallic) * luminance(gb_kd);
    float specularWeight = albedo(specColor, 1.0f, wo, gb_normal);
    float pDiffuse = (diffuseWeight + specularWeight) > 0.f ? diffuseWeight / (diffuseWeight + specularWeight) : 1.f;
    float pSpecular = 1.0f - pDiffuse;
    unsigned int rng_state = hash_pcg(params.rnd_seed, (idx.z * dim.y + idx.y) * dim.x + idx.x);
    unsigned int lightIdx = rand_pcg(rng_state) % params.perms.size(0), bsdfIdx = rand_pcg(rng_state) % params.perms.size(0);
    optixhost::g_idx = idx;
    for (int i = 0; i < (int)(params.n_samples_x * params.n_samples_x); ++i) {
        float sx, sy, sz, pdf_light, pdf_bsdf;
        sx = ((float)(params.perms[lightIdx][i] % params.n_samples_x) + uniform_pcg(rng_state)) * strata_frac;
        sy = ((float)(params.perms[lightIdx][i] / params.n_samples_x) + uniform_pcg(rng_state)) * strata_frac;
        float3 d = lightSample(sx, sy, pdf_light);
        pdf_bsdf = bsdf_pdf(pDiffuse, pSpecular, gb_normal, wo, d, alpha);
        float* o = out + (2 * i) * 6;
        o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = pdf_light; o[4] = pdf_bsdf; o[5] = shadow_test(idx, ray_origin, d, 0.f);
        sx = ((float)(params.perms[bsdfIdx][i] % params.n_samples_x) + uniform_pcg(rng_state)) * strata_frac;
        sy = ((float)(params.perms[bsdfIdx][i] / params.n_samples_x) + uniform_pcg(rng_state)) * strata_frac;
        sz = uniform_pcg(rng_state);
        d = bsdf_sample(pDiffuse, pSpecular, gb_normal, wo, make_float3(sx, sy, sz), alpha, pdf_bsdf);
        pdf_light = lightPDF(d);
        o = out + (2 * i + 1) * 6;
        o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = pdf_light; o[4] = pdf_bsdf; o[5] = shadow_test(idx, ray_origin, d, 0.f);
    }
    return 0;
}

// the same record for a list of pixels (linear index (z * H + y) * W + x) of the LAST launch: out [n_pix, 2*n*n, 6]
int ref_env_shade_trace_pixels(const long long* pix, long long n_pix, float* out) {
    const uint3 dim = optixhost::g_dim;
    const long long S2 = 2LL * params.n_samples_x * params.n_samples_x;
#pragma omp parallel for schedule(dynamic, 16)
    for (long long i = 0; i < n_pix; ++i) {
        const long long p = pix[i];
        ref_env_shade_trace_pixel((int)(p % dim.x), (int)((p / dim.x) % dim.y), (int)(p / ((long long)dim.x * dim.y)), out + i * S2 * 6);
    }
    return 0;
}

}  // extern "C"
