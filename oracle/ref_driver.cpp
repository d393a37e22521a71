/* TEST INFRASTRUCTURE.  C entry points over the reference's OWN, untouched sources, for the units
 * of the hot path that compile in this image without any stand-in header (the ones that never
 * reach math/vec3.h, i.e. glm):
 *     /root/reference/src/libaten/sampler/cmj.h:9-124       (CMJ: permute, randfloat, cmj, sample2D)
 *     /root/reference/src/libaten/sampler/sampler.cpp:8-28  (initSampler / getRandom, std::mt19937)
 *     /root/reference/src/libaten/math/math.h:18-350        (scalar helpers the integrator leans on)
 *     /root/reference/src/libaten/math/vec2.h               (CMJ's return type)
 * Built by `make -C oracle _ref` into oracle/_ref/libatenref.so (git-ignored); this file holds no
 * reference code, only calls.  tests/test_ref_pin.py checks oracle/ == this library bit for bit and
 * tests/golden/make_ref_golden.py mints tests/golden/ref_golden.npz from it.
 */
#include <cstdint>
#include <cstring>

#include "sampler/sampler.h"
#include "math/math.h"

extern "C" {

void ref_init_sampler(uint32_t* out, int32_t w, int32_t h, int32_t seed)
{
    aten::initSampler(w, h, seed);
    const auto& r = aten::getRandom();
    std::memcpy(out, r.data(), r.size() * sizeof(uint32_t));
}

uint32_t ref_get_random(uint32_t idx) { return aten::getRandom(idx); }

/* n successive nextSample() after init(index, dimension, scramble) */
void ref_cmj_samples(uint32_t index, uint32_t dimension, uint32_t scramble, int32_t n, float* out)
{
    aten::CMJ c;
    c.init(index, dimension, scramble);
    for (int32_t i = 0; i < n; i++) out[i] = c.nextSample();
}

/* n successive nextSample2D(): out[2i] = x, out[2i+1] = y */
void ref_cmj_samples2d(uint32_t index, uint32_t dimension, uint32_t scramble, int32_t n, float* out)
{
    aten::CMJ c;
    c.init(index, dimension, scramble);
    for (int32_t i = 0; i < n; i++) {
        const aten::vec2 v = c.nextSample2D();
        out[2 * i] = v.x;
        out[2 * i + 1] = v.y;
    }
}

/* per triple k: init(idx[k], dim[k], scr[k]) then `draws` x nextSample() */
void ref_cmj_batch(int32_t n, const uint32_t* idx, const uint32_t* dim, const uint32_t* scr, int32_t draws, float* out)
{
    for (int32_t k = 0; k < n; k++) {
        aten::CMJ c;
        c.init(idx[k], dim[k], scr[k]);
        for (int32_t d = 0; d < draws; d++) out[(size_t)k * draws + d] = c.nextSample();
    }
}

/* scalar helpers of math/math.h; kinds are shared with orc_math_kat (oracle/aten_oracle.cpp) */
void ref_math_kat(int32_t kind, int32_t n, const float* a, const float* b, const float* c, float* out)
{
    for (int32_t i = 0; i < n; i++) {
        switch (kind) {
        case 0: out[i] = aten::max(a[i], b[i]); break;
        case 1: out[i] = aten::min(a[i], b[i]); break;
        case 2: out[i] = aten::clamp(a[i], b[i], c[i]); break;
        case 3: out[i] = aten::saturate(a[i]); break;
        case 4: out[i] = aten::sign(a[i]); break;
        case 5: out[i] = aten::mix(a[i], b[i], c[i]); break;
        case 6: out[i] = aten::lerp(a[i], b[i], c[i]); break;
        case 7: out[i] = aten::isClose(a[i], b[i], (int32_t)2500) ? 1.0F : 0.0F; break;
        case 8: out[i] = aten::isInvalid(a[i]) ? 1.0F : 0.0F; break;
        case 9: out[i] = aten::sqr(a[i]); break;
        case 10: out[i] = aten::rsqrt(a[i]); break;
        case 11: out[i] = aten::Deg2Rad(a[i]); break;
        default: out[i] = 0.0F;
        }
    }
}

} /* extern "C" */
