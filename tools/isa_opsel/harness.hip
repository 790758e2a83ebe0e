// Host side of tools/isa_opsel/make_variants.py: loads code objects of the old tokred_narrow_kernel<6, true>, runs each four times on
// seeded data and reports, per 16-channel block, run-to-run differences and the distance from an fp64 host reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
int main(int argc, char** argv) {
    const int ldsb = getenv("LDSB") ? atoi(getenv("LDSB")) : 57344;
    const int grid = getenv("GRID") ? atoi(getenv("GRID")) : 512;
    if (grid < 1 || grid > 512) { printf("GRID must be 1..512 (the slab has 512 slices)\n"); return 2; }
    const int C = 96, frames = 128, rpf = 96 * 96; const long P = (long)frames * rpf, tiles = P / 32;
    std::vector<unsigned short> wide((size_t)P * C), narrow((size_t)P * 16);
    std::vector<float> sc((size_t)frames * C), sh((size_t)frames * C);
    srand(1);
    auto rnd = []() { float s = 0; for (int i = 0; i < 4; ++i) s += (rand() % 2001 - 1000) / 1000.f; return s * 0.87f; };
    for (auto& v : wide) v = f2bf(rnd());
    for (auto& v : narrow) v = f2bf(0.01f * rnd());
    for (auto& v : sc) v = 1.f + 0.1f * rnd();
    for (auto& v : sh) v = 0.1f * rnd();
    void *dw, *dn; float *dsc, *dsh, *slab;
    hipMalloc(&dw, wide.size() * 2); hipMalloc(&dn, narrow.size() * 2); hipMalloc(&dsc, sc.size() * 4); hipMalloc(&dsh, sh.size() * 4); hipMalloc(&slab, 512 * C * 16 * 4);
    hipMemcpy(dw, wide.data(), wide.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dn, narrow.data(), narrow.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsh, sh.data(), sh.size() * 4, hipMemcpyHostToDevice);
    // host reference (double, erf GELU) on a subset of columns: channel c, narrow column 0..15
    auto bf2f = [](unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
    std::vector<double> ref(C * 16, 0.0);
    for (long r = 0; r < P; ++r) { int f = (int)(r / rpf); const unsigned short* w = &wide[(size_t)r * C]; const unsigned short* nr = &narrow[(size_t)r * 16];
        float nv[16]; for (int n = 0; n < 16; ++n) nv[n] = bf2f(nr[n]);
        for (int c = 0; c < C; ++c) { double v = (double)bf2f(w[c]) * sc[(size_t)f * C + c] + sh[(size_t)f * C + c]; double g = 0.5 * v * (1.0 + erf(v * 0.70710678118654752));
            for (int n = 0; n < 16; ++n) ref[c * 16 + n] += g * nv[n]; } }
    double rmax = 0; for (double v : ref) rmax = fmax(rmax, fabs(v));
    printf("reference max |value| %.4g\n", rmax);
    for (int a = 1; a < argc; ++a) {
        hipModule_t mod; hipFunction_t fn;
        if (hipModuleLoad(&mod, argv[a]) != hipSuccess) { printf("load failed %s\n", argv[a]); continue; }
        if (hipModuleGetFunction(&fn, mod, "_ZN12_GLOBAL__N_120tokred_narrow_kernelILi6ELb1EEEvPKDF16bS2_lPfPKfS5_i") != hipSuccess) { printf("no kernel\n"); continue; }
        struct { const void* dy; const void* x; long tiles; float* slab; const float* psc; const float* psh; int tpf; } args = {dw, dn, tiles, slab, dsc, dsh, rpf / 32};
        size_t sz = sizeof(args);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        std::vector<std::vector<double>> outs;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(slab, 0, 512 * C * 16 * 4);
            if (hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, ldsb, 0, nullptr, cfg) != hipSuccess) { printf("launch failed\n"); break; }
            hipDeviceSynchronize();
            std::vector<float> h(512 * C * 16);
            hipMemcpy(h.data(), slab, h.size() * 4, hipMemcpyDeviceToHost);
            std::vector<double> o(C * 16, 0.0);
            for (int r = 0; r < 512; ++r) for (int e = 0; e < C * 16; ++e) o[e] += h[(size_t)r * C * 16 + e];
            outs.push_back(o);
        }
        printf("%s: per 16-channel block max |diff| run0-run1:", argv[a]);
        for (int b = 0; b < 6; ++b) { double m = 0; for (int ch = 16 * b; ch < 16 * b + 16; ++ch) for (int n = 0; n < 16; ++n) m = fmax(m, fabs(outs[0][ch * 16 + n] - outs[1][ch * 16 + n])); printf(" %.4g", m); }
        printf("\n   max |run0 - reference| per block:");
        for (int b = 0; b < 6; ++b) { double m = 0; for (int ch = 16 * b; ch < 16 * b + 16; ++ch) for (int n = 0; n < 16; ++n) m = fmax(m, fabs(outs[0][ch * 16 + n] - ref[ch * 16 + n])); printf(" %.4g", m); }
        printf("\n   runs differing from run0 (of 3): "); int nd = 0; for (int r = 1; r < 4; ++r) nd += outs[r] != outs[0]; printf("%d\n", nd);
        hipModuleUnload(mod);
    }
    return 0;
}
