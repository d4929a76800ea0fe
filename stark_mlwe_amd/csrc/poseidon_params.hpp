// stark_mlwe_amd/csrc/poseidon_params.hpp — device view of the Poseidon constants (kernel form).
#pragma once
#include "fr.hpp"
namespace stark {
struct PoseidonDev {           // device pointers to kernel-form constants (see host_util.hpp KernelConsts)
    int t, rf, rp;
    const fr_t* rc_full;       // rf*t
    const fr_t* rc_partial;    // rp
    const fr_t* lu;            // t*t  LU(M)
    const fr_t* lu_pre;        // t*t  LU(B_1*M)
    const fr_t* row0;          // t    M[0][*]
    const fr_t* sparse;        // rp*(2t-1)
    const fr_t* mds;           // t*t  reference form
    const fr_t* mds_pre;       // t*t  dense B_1*M
    const fr_t* gamma;         // (rp/4)*6  cross terms of the 4-round partial blocks
    // the same multiplier tables in radix 2^29, R' = 2^261 domain, 9 words per entry (fr29.hpp): what the dot products read
    const uint32_t* lu29; const uint32_t* lu_pre29; const uint32_t* row0_29; const uint32_t* sparse29; const uint32_t* gamma29;
    const uint32_t* mds29; const uint32_t* mds_pre29;   // dense M and B_1*M (one-wave kernel)
    const void* mds_frag; const void* mds_pre_frag;     // t = 17: the same two matrices as int8 MFMA fragments (host_util.hpp mfma_frags); nullptr otherwise
    // t = 17: the partial rounds unrolled over all rp rounds for the five-wave latency kernel (host_util.hpp chain_*, poseidon_chain.hpp); nullptr otherwise
    const uint32_t* chain_a; const uint32_t* chain_g; const uint32_t* chain_w;
};

}  // namespace stark
