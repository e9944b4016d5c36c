// fft_core.h -- per-thread building blocks of the fused
//   u8-IQ unpack -> (-1)^n centring -> window -> FFT -> |X|^2 accumulate
// kernel (rpf_kernels.hip).  Everything here is plain C++17 on registers and
// pointers so that the same code is compiled (a) by hipcc into the gfx950 kernel
// and (b) by g++ into the thread-by-thread emulator under tests/ that checks
// the index maps on a machine without a GPU.  There is no CPU product path.
//
// What is computed (reference: /root/reference/src/datastore.cxx:66-89):
//   x[n]  = ((float)I_n - 127, (float)Q_n - 127) * (-1)^n [* window[n]]   (:73-77)
//   X[k]  = sum_n x[n] exp(-2 pi i n k / N)                              (:82)
//   pwr[k] += (double)Re^2 + (double)Im^2                                 (:83-85)
//
// Algorithm: decimation-in-frequency FFT, N = P^(NPASS-1) * RLAST, executed by
// T = N/P threads that each hold P complex points in registers.  Passes
// 1..NPASS-1 are radix-P butterflies on elements L_j apart followed by the
// twiddle W_{L_{j-1}}^{m r}; the last pass is P/RLAST radix-RLAST butterflies
// on consecutive elements.  The transform is done "in place" by element name
// e in [0,N): a pass never renames elements, only the LDS exchanges between
// passes move them between threads.  The result is therefore left in
// digit-reversed order, which costs nothing here because the only consumer is
// a per-bin accumulator that lives wherever its bin lands (bin_of()).
#pragma once

#include <cstdint>
#include <type_traits>

#if defined(__HIPCC__)
#define RPF_HD __host__ __device__ __forceinline__
#else
#define RPF_HD inline __attribute__((always_inline))
#endif

namespace rpf {

// A complex float is one 64-bit register pair (re, im): complex add/sub and the
// real-scalar products are single packed-f32 instructions (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32), and the lane swaps / sign flips that complex
// arithmetic needs (multiply by -i, the cross terms of a complex product) ride
// on the VOP3P op_sel / neg modifiers of those instructions instead of costing
// moves.  hipcc does not form those modifier patterns from C++ (it emits
// v_mov/v_xor pairs), so the three primitives below are one-instruction inline
// asm on the device and plain C++ in the host emulator.
typedef float cf __attribute__((ext_vector_type(2)));

// d.lo = (+-)a[S0L] + (+-)b[S1L],  d.hi = (+-)a[S0H] + (+-)b[S1H]
// (S* selects the .x (0) or .y (1) half of the source, N* negates it).
template <int S0L, int S0H, int N0L, int N0H, int S1L, int S1H, int N1L, int N1H>
RPF_HD cf pk_add_mod(cf a, cf b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    cf d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6] neg_lo:[%7,%8] neg_hi:[%9,%10]"
        : "=v"(d)
        : "v"(a), "v"(b), "i"(S0L), "i"(S1L), "i"(S0H), "i"(S1H), "i"(N0L), "i"(N1L), "i"(N0H),
          "i"(N1H));
    return d;
#else
    const float al = S0L ? a.y : a.x, ah = S0H ? a.y : a.x;
    const float bl = S1L ? b.y : b.x, bh = S1H ? b.y : b.x;
    return cf{(N0L ? -al : al) + (N1L ? -bl : bl), (N0H ? -ah : ah) + (N1H ? -bh : bh)};
#endif
}

// a + (-i) b = (a.x + b.y, a.y - b.x)
RPF_HD cf add_mi(cf a, cf b) { return pk_add_mod<0, 1, 0, 0, 1, 0, 0, 1>(a, b); }
// a - (-i) b = (a.x - b.y, a.y + b.x)
RPF_HD cf sub_mi(cf a, cf b) { return pk_add_mod<0, 1, 0, 0, 1, 0, 1, 0>(a, b); }
// -i a = (a.y, -a.x)   and   i a = (-a.y, a.x)
RPF_HD cf mul_mi(cf a) { return pk_add_mod<1, 0, 0, 1, 0, 0, 0, 0>(a, cf{0.0f, 0.0f}); }
RPF_HD cf mul_pi(cf a) { return pk_add_mod<1, 0, 1, 0, 0, 0, 0, 0>(a, cf{0.0f, 0.0f}); }

// complex product a*w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x): two instructions
RPF_HD cf cmul(cf a, cf w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // One asm statement for both instructions: hipcc pads every asm boundary whose
    // result feeds the next instruction with an s_nop, and an s_nop costs the wave
    // an issue slot.  d = (a.x w.x, a.y w.x); then d = (a.y (-w.y) + d.x, a.x w.y + d.y).
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(d)
        : "v"(a), "v"(w));
    return d;
#else
    return cf{__builtin_fmaf(-a.y, w.y, a.x * w.x), __builtin_fmaf(a.x, w.y, a.y * w.x)};
#endif
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }

// ---------------------------------------------------------------- geometry --
template <int N_, int P_>
struct Geom {
    static constexpr int N = N_;
    static constexpr int P = P_;                       // points per thread
    static constexpr int T = N / P;                    // threads per frame
    static constexpr int LOG2N = ilog2(N);
    static constexpr int LOG2P = ilog2(P);
    static constexpr int NPASS = (LOG2N + LOG2P - 1) / LOG2P;
    static constexpr int RLAST = N / ipow(P, NPASS - 1);   // radix of the last pass
    static constexpr int LDS_CPX = N + N / P;           // padded complex slots per frame
    static_assert((1 << LOG2N) == N && (1 << LOG2P) == P, "N and P must be powers of two");
    static_assert(NPASS >= 2, "need at least two passes (N > P)");
    static_assert(T % 8 == 0, "T even: (-1)^n is a per-thread constant; T % 8: 16-byte raw pieces");
    static_assert(P % 8 == 0, "a thread stages 2P raw bytes in 16-byte pieces");

    // sub-FFT length entering pass J (1-based): L_{J-1}
    static constexpr int Lprev(int J) { return N / ipow(P, J - 1); }
    // distance between the P inputs of a pass-J butterfly: L_J
    static constexpr int Lcur(int J) { return Lprev(J) / P; }
    // padded LDS slot of element e: one spare slot after every P elements keeps
    // every access pattern used below free of bank conflicts.
    static RPF_HD int slot(int e) { return e + (e >> LOG2P); }
};

// element held in register a by thread t during pass J
template <class G, int J>
RPF_HD int elem_of(int t, int a)
{
    if constexpr (J < G::NPASS) {
        constexpr int Lp = G::Lprev(J), Lc = G::Lcur(J);
        const int s = t / Lc, m = t % Lc;
        return s * Lp + m + Lc * a;
    } else {
        return G::P * t + a;
    }
}

// Spectrum bin of the value that ends the last pass in register a of thread t.
// Element e = sum_j d_j L_j (digits d_1..d_{NPASS-1} in [0,P), last digit in
// [0,RLAST)) holds X[d_1 + P d_2 + P^2 d_3 + ...].
template <class G, int J = 1>
RPF_HD int bin_digits(int e, int weight)
{
    // digits from the most significant (d_1, weight 1) down; compile-time pass
    // recursion so that every divisor is a constant power of two
    if constexpr (J < G::NPASS) {
        constexpr int L = G::N / ipow(G::P, J);     // L_J
        const int d = e / L;
        return d * weight + bin_digits<G, J + 1>(e - d * L, weight * G::P);
    } else {
        return e * weight;
    }
}
template <class G>
RPF_HD int bin_of(int t, int a)
{
    return bin_digits<G>(G::P * t + a, 1);
}

// ------------------------------------------------------- constant twiddles --
// W_16^k for k in [0,16): (cos, -sin) of k/16 turn, correctly rounded floats.
constexpr float kC1 = 0.92387953251128673848f;   // cos(pi/8)
constexpr float kS1 = 0.38268343236508978178f;   // sin(pi/8)
constexpr float kH = 0.70710678118654752440f;    // sqrt(1/2)

template <int K16>   // multiply by W_16^{K16}
RPF_HD cf mul_w16(cf a)
{
    constexpr int k = ((K16 % 16) + 16) % 16;
    if constexpr (k == 0) return a;
    else if constexpr (k == 4) return mul_mi(a);
    else if constexpr (k == 8) return -a;
    else if constexpr (k == 12) return mul_pi(a);
    // odd multiples of 45 degrees: W = +-H (1 -+ i), one swizzled add and one scale
    else if constexpr (k == 2) return add_mi(a, a) * kH;
    else if constexpr (k == 6) return sub_mi(a, a) * -kH;
    else if constexpr (k == 10) return add_mi(a, a) * -kH;
    else if constexpr (k == 14) return sub_mi(a, a) * kH;
    else {
        // odd k: W = (c, -s) with c,s from {C1,S1} and signs by quadrant
        constexpr float c = (k == 1 || k == 15) ? kC1 : (k == 3 || k == 13) ? kS1
                          : (k == 5 || k == 11) ? -kS1 : -kC1;            // cos(2 pi k/16)
        constexpr float s = (k == 1 || k == 7) ? kS1 : (k == 3 || k == 5) ? kC1
                          : (k == 9 || k == 15) ? -kS1 : -kC1;            // sin(2 pi k/16)
        return cmul(a, cf{c, -s});
    }
}

// ----------------------------------------------------- in-register DFTs --
// dft<R>(v): v[r] <- sum_a v[a] W_R^{a r}, natural order in, natural order out.
template <int R>
struct Dft;

template <>
struct Dft<2> {
    static RPF_HD void run(cf* v)
    {
        const cf a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <>
struct Dft<4> {
    static RPF_HD void run(cf* v)
    {
        const cf apc = v[0] + v[2], amc = v[0] - v[2];
        const cf bpd = v[1] + v[3], bmd = v[1] - v[3];
        v[0] = apc + bpd;
        v[1] = add_mi(amc, bmd);   // amc - i bmd
        v[2] = apc - bpd;
        v[3] = sub_mi(amc, bmd);   // amc + i bmd
    }
};

// R = R1*R2 by one Cooley-Tukey step: a = a0 + R1 a1, r = r0 + R2 r1.
template <int R, int R1, int R2>
struct DftCompose {
    static RPF_HD void run(cf* v)
    {
        static_assert(R == R1 * R2 && 16 % R == 0, "");
        cf col[R2];
#pragma unroll
        for (int a0 = 0; a0 < R1; ++a0) {
#pragma unroll
            for (int a1 = 0; a1 < R2; ++a1) col[a1] = v[a0 + R1 * a1];
            Dft<R2>::run(col);
#pragma unroll
            for (int r0 = 0; r0 < R2; ++r0) v[a0 + R1 * r0] = col[r0];
        }
        twiddle_rows(v, std::integral_constant<int, 0>{});
        cf out[R];
        cf row[R1];
#pragma unroll
        for (int r0 = 0; r0 < R2; ++r0) {
#pragma unroll
            for (int a0 = 0; a0 < R1; ++a0) row[a0] = v[a0 + R1 * r0];
            Dft<R1>::run(row);
#pragma unroll
            for (int r1 = 0; r1 < R1; ++r1) out[r0 + R2 * r1] = row[r1];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = out[r];
    }

    // v[a0 + R1 r0] *= W_R^{a0 r0}; compile-time recursion over idx = a0 + R1 r0
    template <int IDX>
    static RPF_HD void twiddle_rows(cf* v, std::integral_constant<int, IDX>)
    {
        if constexpr (IDX < R) {
            constexpr int a0 = IDX % R1, r0 = IDX / R1;
            v[IDX] = mul_w16<(a0 * r0) * (16 / R)>(v[IDX]);
            twiddle_rows(v, std::integral_constant<int, IDX + 1>{});
        }
    }
};

template <>
struct Dft<8> {
    static RPF_HD void run(cf* v) { DftCompose<8, 2, 4>::run(v); }
};
template <>
struct Dft<16> {
    static RPF_HD void run(cf* v) { DftCompose<16, 4, 4>::run(v); }
};

// ------------------------------------------------------------------ phases --
// u8 -> f32 without a (half-rate) v_cvt: OR the byte into the mantissa of 2^23,
// i.e. bits 0x4B0000bb are the float 8388608 + b exactly.
RPF_HD float byte_plus_2p23(uint32_t b)
{
    return __builtin_bit_cast(float, 0x4B000000u | b);
}
// (2^23 + I, 2^23 + Q) of one interleaved sample iq = I | Q << 8: on the device
// one v_perm_b32 per component picks the byte and the 0x4B exponent byte at once.
RPF_HD cf iq_plus_2p23(uint32_t iq)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // perm(S0, S1, sel): result byte i = byte sel[i] of {S0 (4..7), S1 (0..3)}; 0x0c = zero
    const uint32_t bi = __builtin_amdgcn_perm(0x4B000000u, iq, 0x070c0c00u);
    const uint32_t bq = __builtin_amdgcn_perm(0x4B000000u, iq, 0x070c0c01u);
    return cf{__builtin_bit_cast(float, bi), __builtin_bit_cast(float, bq)};
#else
    return cf{byte_plus_2p23(iq & 0xffu), byte_plus_2p23((iq >> 8) & 0xffu)};
#endif
}
// the same for sample h (0 or 1) of a register that holds two interleaved samples, I0 Q0 I1 Q1
template <int H>
RPF_HD cf iq_pair_plus_2p23(uint32_t iq2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t bi = __builtin_amdgcn_perm(0x4B000000u, iq2, H ? 0x070c0c02u : 0x070c0c00u);
    const uint32_t bq = __builtin_amdgcn_perm(0x4B000000u, iq2, H ? 0x070c0c03u : 0x070c0c01u);
    return cf{__builtin_bit_cast(float, bi), __builtin_bit_cast(float, bq)};
#else
    return iq_plus_2p23(H ? iq2 >> 16 : iq2 & 0xffffu);
#endif
}
constexpr float kTwo23 = 8388608.0f;

// Raw-byte staging is wavefront-local: a wave stages exactly the samples its own
// 64 threads unpack, "a-major": bytes [128 a, 128 a + 128) of the wave's 128*P-byte
// raw area hold sample n = t + T a of its 64 threads, so lane l reads its P samples
// at 2 l + 128 a and no workgroup barrier is needed between staging and unpack.
// raw_source maps byte j of that area back to the stream: which of the
// workgroup's frame slots, and which byte of that frame.  16-byte pieces (j % 16
// == 0) are contiguous and 16-byte aligned in the source because T % 8 == 0.
constexpr int kRawChunk = 128;
template <class G>
RPF_HD void raw_source(int wave_in_wg, int j, int* slot, int* byte_in_frame)
{
    const int a = j / kRawChunk;
    const int tid = wave_in_wg * 64 + ((j % kRawChunk) >> 1);
    *slot = tid / G::T;
    *byte_in_frame = 2 * (tid % G::T + G::T * a) + (j & 1);
}

// Unpack the P samples of one thread (datastore.cxx:73-77).  `lane_raw` points at
// this lane's column of its wave's raw area.  sgn = (-1)^t (n = t + T a, T even).
// wsgn: per-register window values already multiplied by sgn (WINDOW only).
template <class G, bool WINDOW>
RPF_HD void phase_unpack(const uint8_t* lane_raw, float sgn, const float* wsgn, cf* x)
{
    // (2^23 + v) * sgn - (2^23 + 127) * sgn = (v - 127) * sgn, every step exact
    const float off = -(kTwo23 + 127.0f) * sgn;
#pragma unroll
    for (int a = 0; a < G::P; ++a) {
        const uint32_t iq = *reinterpret_cast<const uint16_t*>(lane_raw + kRawChunk * a);
        const cf f = iq_plus_2p23(iq);
        if constexpr (WINDOW) {
            // (v - 127) is exact, * (+-w) rounds once: same value as the reference
            x[a] = (f - (kTwo23 + 127.0f)) * wsgn[a];
        } else {
            x[a] = f * sgn + off;
        }
    }
}

// Radix-P butterfly of a middle pass followed by its twiddles tw[r-1] =
// W_{L_{J-1}}^{m r}, r = 1..P-1 (m = t mod L_J; loaded once per thread).
template <class G>
RPF_HD void phase_butterfly_twiddle(cf* x, const cf* tw)
{
    Dft<G::P>::run(x);
#pragma unroll
    for (int r = 1; r < G::P; ++r) x[r] = cmul(x[r], tw[r - 1]);
}

// Last pass: P/RLAST radix-RLAST butterflies on consecutive registers.
template <class G>
RPF_HD void phase_last(cf* x)
{
#pragma unroll
    for (int g = 0; g < G::P / G::RLAST; ++g) Dft<G::RLAST>::run(x + g * G::RLAST);
}

// LDS exchange: registers of pass J <-> padded per-frame slab.  The slot of
// register a is slot_base<J>(t) + slot_delta<J>(a): a per-thread, loop-invariant
// base plus a compile-time constant that folds into the DS instruction's
// immediate offset.  (slot(b + Lc a) = slot(b) + Lc a + floor(Lc a / P) holds
// because Lc a is a multiple of P when Lc >= P, and m < Lc | P otherwise.)
template <class G, int J>
RPF_HD int slot_base(int t)
{
    return G::slot(elem_of<G, J>(t, 0));
}
template <class G, int J>
constexpr int slot_delta(int a)
{
    if constexpr (J < G::NPASS) return G::Lcur(J) * a + (G::Lcur(J) * a) / G::P;
    else return a;
}
template <class G, int J>
RPF_HD void phase_store(int t, const cf* x, cf* slab)
{
    cf* const p = slab + slot_base<G, J>(t);
#pragma unroll
    for (int a = 0; a < G::P; ++a) p[slot_delta<G, J>(a)] = x[a];
}
// On the device the fetch goes through a volatile LDS-address-space pointer only
// to keep hipcc from pairing the loads into ds_read2_b64, which moves half the
// bytes per LDS cycle of two ds_read_b64 (MI355X_MICROARCH.md, LDS table).
// RPF_LDS_PLAIN_FETCH builds the plain-pointer variant for A/B measurement.
template <class G, int J>
RPF_HD void phase_fetch(int t, cf* x, const cf* slab)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RPF_LDS_PLAIN_FETCH)
    using lds_cf = const volatile __attribute__((address_space(3))) cf;
    lds_cf* const p = (lds_cf*)(slab + slot_base<G, J>(t));
#else
    const cf* const p = slab + slot_base<G, J>(t);
#endif
#pragma unroll
    for (int a = 0; a < G::P; ++a) x[a] = p[slot_delta<G, J>(a)];
}

// pwr += Re^2 + Im^2 in double (datastore.cxx:83-85).  The reference rounds
// Re^2+Im^2 (both squares exact in double) and then the running sum; here each
// exact square is folded in by its own fused multiply-add -- also two roundings
// per bin and frame, ~1e-16 relative, but two half-rate instructions fewer.
RPF_HD void phase_accumulate(const cf* x, double* acc, int P)
{
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const double re = static_cast<double>(x[a].x);
        const double im = static_cast<double>(x[a].y);
        acc[a] = __builtin_fma(im, im, __builtin_fma(re, re, acc[a]));
    }
}

// Index into the master twiddle table W_N^k (k in [0,N)) of the pass-J twiddle
// W_{L_{J-1}}^{m r}:  k = m * r * P^(J-1).
template <class G, int J>
RPF_HD int twiddle_index(int t, int r)
{
    constexpr int Lc = G::Lcur(J);
    return (t % Lc) * r * ipow(G::P, J - 1);
}

}  // namespace rpf
