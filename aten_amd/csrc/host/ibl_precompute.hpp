// Importance-sampling tables of an equirectangular environment map: ImageBasedLight::preCompute
// (src/libaten/light/ibl.cpp:10-118).  Row PDFs are weighted by sin(theta) of the row, the per-row U tables and the V
// table become CDFs normalised to [0, 1].  Pure host C++ (built with -ffp-contract=off like everything else).
#pragma once
#include <cmath>
#include <vector>

namespace atn {

struct IblTables {
    std::vector<float> cdf_v;   // [h]
    std::vector<float> cdf_u;   // [h * w], row-major
    float avg_illum = 0.0F;
    int32_t w = 0, h = 0;
};

// texel(x, y) must return the envmap's colour AT the texel centre as texture::at does for u = (x + 0.5) / w,
// v = (y + 0.5) / h, already multiplied by the background's `multiplyer` (SampleFromUVWithTexture, ibl.h:147-153).
template <class TexelFn>
inline void ibl_precompute(IblTables& t, int32_t width, int32_t height, TexelFn texel)
{
    t.w = width; t.h = height;
    t.cdf_v.assign((size_t)height, 0.0F);
    t.cdf_u.assign((size_t)width * height, 0.0F);
    float avg = 0.0F, total_weight = 0.0F;
    for (int32_t y = 0; y < height; y++) {
        // latitude correction of the equirectangular map; + 0.5 samples the texel centre and keeps the scale non-zero
        const float scale = std::sin(3.14159265358979323846F * (float)(y + 0.5) / height);
        float pdf_v = 0.0F;
        for (int32_t x = 0; x < width; x++) {
            float r, g, b;
            texel(x, y, r, g, b);
            const float illum = (0.212639F * r + 0.71517F * g) + 0.0721926F * b;        // color::luminance
            avg += illum * scale;
            total_weight += scale;
            pdf_v += illum * scale;
            t.cdf_u[(size_t)y * width + x] = illum * scale;
        }
        t.cdf_v[y] = pdf_v;
    }
    auto to_cdf = [](float* c, int32_t n) {
        float sum = 0.0F;
        for (int32_t i = 0; i < n; i++) {
            sum += c[i];
            if (i > 0) c[i] += c[i - 1];
        }
        if (sum > 0.0F) {
            const float inv = 1 / sum;
            for (int32_t i = 0; i < n; i++) {
                c[i] *= inv;
                c[i] = c[i] < 0.0F ? 0.0F : (c[i] > 1.0F ? 1.0F : c[i]);
            }
        }
    };
    to_cdf(t.cdf_v.data(), height);
    for (int32_t y = 0; y < height; y++) to_cdf(t.cdf_u.data() + (size_t)y * width, width);
    t.avg_illum = avg / total_weight;
}

} // namespace atn
