// Correlated multi-jittered sampler, integer-exact restatement of aten::CMJ
// (src/libaten/sampler/cmj.h:9-124).  nextSample() consumes one dimension and returns only
// the x component of the 2-D CMJ point, so only x is computed here.
#pragma once
#include "vec.hpp"

namespace atn {

struct Cmj { uint32_t idx, dim, scramble; };

ATN_DEV uint32_t cmj_permute(uint32_t i, uint32_t l, uint32_t p)   // cmj.h:51-85
{
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p;             i *= 0xe170893d;
        i ^= p >> 16;       i ^= (i & w) >> 4;
        i ^= p >> 8;        i *= 0x0929eb3f;
        i ^= p >> 23;       i ^= (i & w) >> 1;
        i *= 1 | p >> 27;   i *= 0x6935fa69;
        i ^= (i & w) >> 11; i *= 0x74dcb303;
        i ^= (i & w) >> 2;  i *= 0x9e501cc3;
        i ^= (i & w) >> 2;  i *= 0xc860a3df;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}

ATN_DEV float cmj_randfloat(uint32_t i, uint32_t p)                 // cmj.h:87-101
{
    i ^= p;
    i ^= i >> 17; i ^= i >> 10; i *= 0xb36534e5;
    i ^= i >> 12; i ^= i >> 21; i *= 0x93fc4795;
    i ^= 0xdf6e307f;
    i ^= i >> 17; i *= 1 | p >> 18;
    return (float)i * (1.0f / 4294967808.0f);
}

ATN_DEV float cmj_next(Cmj& s)                                      // cmj.h:32-37,103-121
{
    constexpr int32_t n = 16;   // CMJ_DIM
    const uint32_t ds = s.dim * s.scramble;
    const int32_t k = (int32_t)cmj_permute(s.idx, n * n, 0xa399d265u * s.dim * s.scramble);
    const uint32_t p = ds;
    const int32_t sy = (int32_t)cmj_permute((uint32_t)(k / n), n, p * 0x63d83595u);
    const float jx = cmj_randfloat((uint32_t)k, p * 0xa399d265u);
    s.dim++;
    return ((float)(k % n) + ((float)sy + jx) / (float)n) / (float)n;
}

} // namespace atn
