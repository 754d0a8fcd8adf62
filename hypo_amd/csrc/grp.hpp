// grp.hpp — "group" primitives of the POA kernel.
//
// A group is GW lanes (16/32/64) of one wavefront that together own one window.  Control flow is
// group-uniform: every lane of a group executes every collective in the same order.  Lanes exchange
// data through registers (shuffle / ballot) or through the group's private LDS slice, always
// separated by Grp::sync().
//
// Two back ends:
//   * device (hipcc, gfx950): wave64 cross-lane instructions.  GW == 64 maps to full-wave ops.
//   * HYPO_EMU (g++, tests only): every lane is a fiber scheduled round-robin; a collective is a
//     rendezvous through a per-group mailbox.  This lets tests/ run the *same* poa_core.hpp on the
//     CPU (under -fsanitize=address,undefined) against the oracle before a GPU is involved.
#pragma once
#include <stdint.h>

#ifdef HYPO_EMU
#define HD inline
#define HD_NOINLINE inline
#define HYPO_UNROLL
#define HYPO_NOUNROLL
#define HYPO_IN_VGPR(x) do { } while (0)
#define HYPO_NO_IFCVT() do { } while (0)
#define HYPO_ARRIVED(x) do { } while (0)
#else
#include <hip/hip_runtime.h>
#define HD __device__ __forceinline__
// a real call: the callee's registers do not add to what the (huge) caller keeps alive
#define HD_NOINLINE __device__ __attribute__((noinline))
#define HYPO_UNROLL _Pragma("unroll")
// cold loops (spill / restore of a re-queued window's graph): unrolled copies of them would set the kernel's register count
#define HYPO_NOUNROLL _Pragma("clang loop unroll(disable)")
// keeps a group-uniform value in a vector register (the row loop runs out of scalar registers and the compiler would
// otherwise park it in a VGPR lane and v_readlane it back on every use)
#define HYPO_IN_VGPR(x) asm volatile("" : "+v"(x))
// an empty volatile statement keeps a rarely taken, group-uniform branch a branch (no if-conversion into selects)
#define HYPO_NO_IFCVT() asm volatile("")
// consumes a loaded value here, so the wait for the load is placed here and not at the first use inside a loop (where it
// would be an s_waitcnt lgkmcnt(0) per iteration that also waits for the previous iteration's LDS stores)
#define HYPO_ARRIVED(x) asm volatile("" :: "v"(x))
#endif

// Device-coherent accesses (agent scope, relaxed): what one wave hands to a wave that may run on another XCD WHILE BOTH RUN — a re-queued
// window's spill and carry word.  On gfx942 / gfx950 such a store is written through its XCD's L2 and such a load is served from beyond
// it (the sc1 bit), so neither side needs an agent-scope fence: a release fence at that scope writes back the whole L2 of the XCD
// (buffer_wbl2) and an acquire invalidates it (buffer_inv) — measured at tens of microseconds each, serialised per XCD, once per
// re-queued window: a batch in which a fifth of the windows outgrew their class spent 160 of its 162 ms there (round 6,
// profiles/r06_grid_diag.txt).  Ordering against the queue entry that publishes the window is a plain wait for the stores (workgroup-scope
// release).  The emulator reads and writes memory.
#ifdef HYPO_EMU
#define HYPO_ST_DEV(p, v) (*(p) = (v))
#define HYPO_LD_DEV(p) (*(p))
#define HYPO_RELEASE_STORES() do { } while (0)
#else
#define HYPO_ST_DEV(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define HYPO_LD_DEV(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define HYPO_RELEASE_STORES() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#endif

namespace hypo {

struct alignas(16) uint4v { uint32_t x, y, z, w; };       // 16-byte store unit

#ifdef HYPO_EMU
// ---- emulator back end -------------------------------------------------------------------------
struct EmuGroup {              // shared by the GW fibers of one group
    int gw;
    int64_t box[64];           // mailbox
    void (*yield)(void*);      // switch to the next lane's fiber
    void* sched;
};

template <int GW>
struct Grp {
    int lane;
    EmuGroup* eg;
    HD void sync() const { eg->yield(eg->sched); }
    template <class T> HD T shfl(T v, int src) const {
        eg->box[lane] = (int64_t)v; sync();
        T r = (T)eg->box[src & (GW - 1)]; sync();
        return r;
    }
    // value of lane-1 (lane 0 gets `fill`)
    template <class T> HD T shfl_up1(T v, T fill) const {
        eg->box[lane] = (int64_t)v; sync();
        T r = lane ? (T)eg->box[lane - 1] : fill; sync();
        return r;
    }
    HD uint64_t ballot(bool p) const {
        eg->box[lane] = p ? 1 : 0; sync();
        uint64_t m = 0;
        for (int i = 0; i < GW; ++i) m |= (uint64_t)(eg->box[i] & 1) << i;
        sync();
        return m;
    }
    HD bool any(bool p) const { return ballot(p) != 0; }
    HD int reduce_max(int v) const {
        eg->box[lane] = v; sync();
        int r = (int)eg->box[0];
        for (int i = 1; i < GW; ++i) if ((int)eg->box[i] > r) r = (int)eg->box[i];
        sync();
        return r;
    }
    HD int reduce_add(int v) const {
        eg->box[lane] = v; sync();
        int r = 0;
        for (int i = 0; i < GW; ++i) r += (int)eg->box[i];
        sync();
        return r;
    }
    // exclusive prefix max over lanes (lane 0 gets `ident`)
    HD int scan_max_excl(int v, int ident) const {
        eg->box[lane] = v; sync();
        int r = ident;
        for (int i = 0; i < lane; ++i) if ((int)eg->box[i] > r) r = (int)eg->box[i];
        sync();
        return r;
    }
    // same with the identity handed in as a loop-carried value (device: lane 0 of the final shift keeps what it had)
    HD int scan_max_excl_c(int v, int carry) const { const int r = scan_max_excl(v, (int)0x80000000); return lane ? r : carry; }
    // inclusive prefix sum over the group's lanes
    HD int scan_add_incl(int v) const {
        eg->box[lane] = v; sync();
        int r = 0;
        for (int i = 0; i <= lane; ++i) r += (int)eg->box[i];
        sync();
        return r;
    }
    HD int uniform(int v) const { return v; }
};
HD int popc64(uint64_t x) { return __builtin_popcountll(x); }
HD int ctz64(uint64_t x) { return __builtin_ctzll(x); }
HD int clz64(uint64_t x) { return __builtin_clzll(x); }

#else
// ---- device back end (gfx950, wave64) ------------------------------------------------------------
// Cross-lane traffic uses DPP (row_shr / row_bcast / wave_shr: register-to-register, a few cycles) and
// v_readlane for group-uniform indices; ds_bpermute (LDS pipe, ~100 cycles) only for per-lane indices.
template <int CTRL, int ROW_MASK = 0xf>
HD int dpp_mov(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }

template <int GW>
struct Grp {
    int lane;                                   // lane inside the group
    HD int wlane() const { return (int)(threadIdx.x & 63); }
    HD int gbase() const { return wlane() & ~(GW - 1); }
    // LDS traffic of one wave is processed in order; this only stops the compiler from moving
    // memory operations across the rendezvous (wavefront-scope fences emit no instructions).
    HD void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // src must be group-uniform
    template <class T> HD T shfl(T v, int src) const {
        if (GW == 64) return (T)__builtin_amdgcn_readlane((int)v, src);
        return (T)__shfl((int)v, src, GW);
    }
    template <class T> HD T shfl_up1(T v, T fill) const {
        int r;
        if (GW == 16) r = dpp_mov<0x111>((int)fill, (int)v);             // row_shr:1
        else r = dpp_mov<0x138>((int)fill, (int)v);                      // wave_shr:1
        if (GW == 32) r = lane ? r : (int)fill;
        return (T)r;
    }
    HD uint64_t ballot(bool p) const {
        uint64_t m = __ballot(p);
        if (GW == 64) return m;
        return (m >> gbase()) & ((1ull << (GW & 63)) - 1ull);
    }
    HD bool any(bool p) const { return ballot(p) != 0; }
    // Reductions over the group: an inclusive DPP scan (register to register, six dependent VALU operations) whose last lane holds
    // the result, handed to the group by v_readlane (64-lane groups) or one ds_bpermute.  (Rounds 1-4 folded with __shfl_xor:
    // five or six ds_bpermute round trips through the LDS crossbar per reduction, ~600 cycles each time a phase asked.)
    HD int reduce_max(int v) const {
        constexpr int ID = (int)0x80000000;
        int x = v, t;
        t = dpp_mov<0x111>(ID, x); x = t > x ? t : x;                 // row_shr:1
        t = dpp_mov<0x112>(ID, x); x = t > x ? t : x;                 // row_shr:2
        t = dpp_mov<0x114>(ID, x); x = t > x ? t : x;                 // row_shr:4
        t = dpp_mov<0x118>(ID, x); x = t > x ? t : x;                 // row_shr:8
        if (GW >= 32) { t = dpp_mov<0x142, 0xa>(ID, x); x = t > x ? t : x; }   // row_bcast:15 -> rows 1,3
        if (GW == 64) { t = dpp_mov<0x143, 0xc>(ID, x); x = t > x ? t : x; }   // row_bcast:31 -> rows 2,3
        if (GW == 64) return __builtin_amdgcn_readlane(x, 63);          // wave-uniform -> scalar control flow
        return __shfl(x, GW - 1, GW);
    }
    HD int reduce_add(int v) const {
        int x = v;
        x += dpp_mov<0x111>(0, x);
        x += dpp_mov<0x112>(0, x);
        x += dpp_mov<0x114>(0, x);
        x += dpp_mov<0x118>(0, x);
        if (GW >= 32) x += dpp_mov<0x142, 0xa>(0, x);
        if (GW == 64) x += dpp_mov<0x143, 0xc>(0, x);
        if (GW == 64) return __builtin_amdgcn_readlane(x, 63);
        return __shfl(x, GW - 1, GW);
    }
    // exclusive prefix max over the group's lanes; lane 0 gets INT_MIN.  INT_MIN is max's identity, which lets
    // the compiler fold every row_shr / row_bcast move into the v_max_i32 that consumes it (one VALU op per step).
    HD int scan_max_excl(int v, int /*ident*/) const {
        constexpr int ID = (int)0x80000000;
        int x = v, t;
        t = dpp_mov<0x111>(ID, x); x = t > x ? t : x;                 // row_shr:1
        t = dpp_mov<0x112>(ID, x); x = t > x ? t : x;                 // row_shr:2
        t = dpp_mov<0x114>(ID, x); x = t > x ? t : x;                 // row_shr:4
        t = dpp_mov<0x118>(ID, x); x = t > x ? t : x;                 // row_shr:8
        if (GW >= 32) { t = dpp_mov<0x142, 0xa>(ID, x); x = t > x ? t : x; }   // row_bcast:15 -> rows 1,3
        if (GW == 64) { t = dpp_mov<0x143, 0xc>(ID, x); x = t > x ? t : x; }   // row_bcast:31 -> rows 2,3
        return shfl_up1(x, ID);
    }
    // Same scan for a loop: `carry` is the previous iteration's result.  The final shift never writes lane 0 of a group, so
    // that lane keeps the identity it started with and no constant has to be re-materialised per iteration.
    HD int scan_max_excl_c(int v, int carry) const {
        constexpr int ID = (int)0x80000000;
        int x = v, t;
        t = dpp_mov<0x111>(ID, x); x = t > x ? t : x;
        t = dpp_mov<0x112>(ID, x); x = t > x ? t : x;
        t = dpp_mov<0x114>(ID, x); x = t > x ? t : x;
        t = dpp_mov<0x118>(ID, x); x = t > x ? t : x;
        if (GW >= 32) { t = dpp_mov<0x142, 0xa>(ID, x); x = t > x ? t : x; }
        if (GW == 64) { t = dpp_mov<0x143, 0xc>(ID, x); x = t > x ? t : x; }
        return shfl_up1(x, carry);
    }
    // inclusive prefix sum over the group's lanes (cold code: window staging)
    HD int scan_add_incl(int v) const {
        HYPO_UNROLL
        for (int d = 1; d < GW; d <<= 1) { const int t = __shfl_up(v, d, GW); if (lane >= d) v += t; }
        return v;
    }
    // hint: value is identical in every lane of the WAVE (only true for GW == 64)
    HD int uniform(int v) const { return GW == 64 ? __builtin_amdgcn_readfirstlane(v) : v; }
};
HD int popc64(uint64_t x) { return __popcll(x); }
HD int ctz64(uint64_t x) { return __ffsll((unsigned long long)x) - 1; }
HD int clz64(uint64_t x) { return __clzll((long long)x); }
#endif

// ---- packed pairs of int16 (two neighbouring DP columns in one register) -------------------------------------
// Device: VOP3P packed math (v_pk_add_u16, v_pk_max_i16, v_pk_min_u16, v_pk_mad_u16), v_alignbit / v_perm to move halves.
// All arithmetic wraps modulo 2^16.  The emulator restates every operation on two scalars.
#ifdef HYPO_EMU
struct P2 { int16_t lo, hi; };
HD P2 pk_make(int lo, int hi) { return P2{(int16_t)lo, (int16_t)hi}; }
HD P2 pk_splat(int x) { return P2{(int16_t)x, (int16_t)x}; }
HD int pk_lo(P2 a) { return a.lo; }
HD int pk_hi(P2 a) { return a.hi; }
HD int pk_bits(P2 a) { return (int)((uint32_t)(uint16_t)a.lo | ((uint32_t)(uint16_t)a.hi << 16)); }
HD P2 pk_from_bits(int b) { return P2{(int16_t)(uint16_t)((uint32_t)b & 0xffffu), (int16_t)(uint16_t)((uint32_t)b >> 16)}; }
HD P2 pk_add(P2 a, P2 b) { return P2{(int16_t)(uint16_t)((uint16_t)a.lo + (uint16_t)b.lo), (int16_t)(uint16_t)((uint16_t)a.hi + (uint16_t)b.hi)}; }
HD P2 pk_sub(P2 a, P2 b) { return P2{(int16_t)(uint16_t)((uint16_t)a.lo - (uint16_t)b.lo), (int16_t)(uint16_t)((uint16_t)a.hi - (uint16_t)b.hi)}; }
HD P2 pk_max(P2 a, P2 b) { return P2{a.lo > b.lo ? a.lo : b.lo, a.hi > b.hi ? a.hi : b.hi}; }
HD P2 pk_minu(P2 a, P2 b) { return P2{(int16_t)((uint16_t)a.lo < (uint16_t)b.lo ? a.lo : b.lo), (int16_t)((uint16_t)a.hi < (uint16_t)b.hi ? a.hi : b.hi)}; }
HD P2 pk_mad(P2 a, P2 b, P2 c) {
    return P2{(int16_t)(uint16_t)((uint32_t)(uint16_t)a.lo * (uint16_t)b.lo + (uint16_t)c.lo), (int16_t)(uint16_t)((uint32_t)(uint16_t)a.hi * (uint16_t)b.hi + (uint16_t)c.hi)};
}
HD P2 pk_xor(P2 a, P2 b) { return pk_from_bits(pk_bits(a) ^ pk_bits(b)); }
// (lo, hi) = (a.hi, b.lo): the pair one column to the left of b, a being the pair before it
HD P2 pk_shift_in(P2 a, P2 b) { return P2{a.hi, b.lo}; }
HD P2 pk_hi_splat(P2 a) { return P2{a.hi, a.hi}; }
// hi = max(hi, lo), lo unchanged
HD P2 pk_fold_hi(P2 a) { return P2{a.lo, a.hi > a.lo ? a.hi : a.lo}; }
#else
typedef short P2 __attribute__((ext_vector_type(2)));
typedef unsigned short P2u __attribute__((ext_vector_type(2)));
HD P2 pk_from_bits(int b) { return __builtin_bit_cast(P2, b); }
HD int pk_bits(P2 a) { return __builtin_bit_cast(int, a); }
HD P2 pk_make(int lo, int hi) { return pk_from_bits((int)(((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16))); }
HD P2 pk_splat(int x) { return pk_make(x, x); }
HD int pk_lo(P2 a) { return (int)a.x; }
HD int pk_hi(P2 a) { return (int)a.y; }
HD P2 pk_add(P2 a, P2 b) { return a + b; }
HD P2 pk_sub(P2 a, P2 b) { return a - b; }
HD P2 pk_max(P2 a, P2 b) { return __builtin_elementwise_max(a, b); }
HD P2 pk_minu(P2 a, P2 b) { return __builtin_bit_cast(P2, __builtin_elementwise_min(__builtin_bit_cast(P2u, a), __builtin_bit_cast(P2u, b))); }
HD P2 pk_mad(P2 a, P2 b, P2 c) { return a * b + c; }
HD P2 pk_xor(P2 a, P2 b) { return pk_from_bits(pk_bits(a) ^ pk_bits(b)); }
HD P2 pk_shift_in(P2 a, P2 b) { return pk_from_bits((int)__builtin_amdgcn_alignbit((uint32_t)pk_bits(b), (uint32_t)pk_bits(a), 16)); }
HD P2 pk_hi_splat(P2 a) { return pk_from_bits((int)__builtin_amdgcn_perm((uint32_t)pk_bits(a), (uint32_t)pk_bits(a), 0x07060706u)); }
HD P2 pk_fold_hi(P2 a) { return pk_max(a, pk_from_bits((int)(((uint32_t)pk_bits(a) << 16) | 0x8000u))); }
#endif

}  // namespace hypo
