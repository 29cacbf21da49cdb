// pkmath.h -- complex arithmetic on packed fp32 instructions (v_pk_add/mul/fma_f32) for gfx950.
//
// A complex value lives in an aligned VGPR pair (re, im).  One packed instruction works on
// both halves, and its op_sel / neg modifiers pick and negate the halves per source, so a
// complex add, a multiplication by +-i folded into an add, a conjugation ... are ONE instruction
// and a complex multiply is two: a radix-16 butterfly with its twiddles is 108 packed + 0 plain
// instructions instead of 214.  The compiler folds broadcasts and swaps into op_sel but not the
// mixed negations, hence the inline assembly (not volatile: it schedules freely).
// Measured (tools/ubench/pk_dft16.hip, profiles/r02x_packed_fp32.txt): no gain at the occupancy
// of the product kernels -- a packed instruction costs two plain ones once a SIMD has >= 2 waves
// -- so the product does NOT use this header.
#pragma once
#include <hip/hip_runtime.h>

namespace setk {

typedef float pk2 __attribute__((ext_vector_type(2)));

#define SETK_PK __device__ __forceinline__

SETK_PK pk2 pk(float2 a) { return (pk2){a.x, a.y}; }
SETK_PK float2 unpk(pk2 a) { return make_float2(a.x, a.y); }

#define SETK_PK_ADD(name, mods)                                                      \
    SETK_PK pk2 name(pk2 a, pk2 b) {                                                 \
        pk2 r;                                                                       \
        asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));             \
        return r;                                                                    \
    }
SETK_PK pk2 pk_add(pk2 a, pk2 b) { return a + b; }
SETK_PK pk2 pk_sub(pk2 a, pk2 b) { return a - b; }
// a + (-i) b = (a.x + b.y, a.y - b.x)       a - (-i) b = a + i b = (a.x - b.y, a.y + b.x)
SETK_PK_ADD(pk_add_mi, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
SETK_PK_ADD(pk_add_pi, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// a + conj(b), a - conj(b)
SETK_PK_ADD(pk_add_conj, "neg_hi:[0,1]")
SETK_PK_ADD(pk_sub_conj, "neg_lo:[0,1]")
// conj(a) - conj(b) ... not needed; (a.y + b.y, b.x - a.x): the "odd" half of the real split
SETK_PK_ADD(pk_split_odd, "op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]")
// (a.x - b.x, b.y - a.y) = conj(a - b)
SETK_PK_ADD(pk_sub_then_conj, "neg_lo:[0,1] neg_hi:[1,0]")
#undef SETK_PK_ADD

// a * w (complex)
SETK_PK pk2 pk_cmul(pk2 a, pk2 w) {
    pk2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w)
SETK_PK pk2 pk_cmulc(pk2 a, pk2 w) {
    pk2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * w with a uniform (scalar-register) w
SETK_PK pk2 pk_cmul_s(pk2 a, pk2 w) {
    pk2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
// a * (s, s)
SETK_PK pk2 pk_scale_s(pk2 a, float s) { return a * (pk2){s, s}; }
// elementwise a * b, fma(a, b, c)
SETK_PK pk2 pk_mul(pk2 a, pk2 b) { return a * b; }
SETK_PK pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
// c + (w.x, w.x) * a   and   c + (w.y, w.y) * a   (w = a pair of real weights)
SETK_PK pk2 pk_fma_lo(pk2 a, pk2 w, pk2 c) {
    pk2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}
SETK_PK pk2 pk_fma_hi(pk2 a, pk2 w, pk2 c) {
    pk2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}

}  // namespace setk
