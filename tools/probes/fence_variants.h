// Variants of the fence / asm-statement hooks of ze_attn_batch.hip for tools/probes/fence_hunt.sh (-include'd, -DFH=<n>).
#pragma once
#if FH == 0 || FH == 11
#define AW_FENCED 1
#endif
#if FH >= 1 && FH <= 9
#define AD_PACK_ASM 1   // variants 1-9 reproduce round 4: the conversion as inline asm
#endif
#define FH_NONE() do {} while (0)
#if FH == 1    // nofence
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#elif FH == 2  // only A (behind the wait + K Q^T statement)
#undef AW_FENCED
#define AW_FENCE_A() __builtin_amdgcn_sched_barrier(0)
#define AW_FENCE_B() FH_NONE()
#elif FH == 3  // only B (behind the P V loop = in front of the next statement)
#undef AW_FENCED
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() __builtin_amdgcn_sched_barrier(0)
#elif FH == 4  // nofence, 64 more wait states behind the statement's MFMAs
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#define AW_ASM_POST "\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
#elif FH == 5  // nofence, 64 wait states between the wait and the first MFMA
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#define AW_ASM_PRE "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#elif FH == 6  // nofence, every LDS read returned before the statement's MFMAs
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#define AW_ASM_PRE "s_waitcnt lgkmcnt(0)\n\t"
#elif FH == 7  // nofence, every LDS read returned behind the statement's MFMAs
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#define AW_ASM_POST "\n\ts_waitcnt lgkmcnt(0)"
#elif FH == 8  // nofence, two wait states BEHIND every v_cvt_pk_bf16_f32 (its result -> the MFMA that reads it)
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#define AD_PACK_POST "\n\ts_nop 1"
#elif FH == 9  // nofence, two wait states IN FRONT of every v_cvt_pk_bf16_f32 (the v_exp_f32 results it reads)
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#define AD_PACK_PRE "s_nop 1\n\t"
#elif FH == 10  // nofence, the conversion left to the compiler (it then knows the instruction and its hazards): what ships
#define AW_FENCE_A() FH_NONE()
#define AW_FENCE_B() FH_NONE()
#elif FH == 11  // fenced + the asm conversion: round 4's shipped build
#define AD_PACK_ASM 1
#endif
