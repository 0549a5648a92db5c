// The sweep loop of lattice_sentence (lattice.hip) for the common build -- i16 connection cells, sentences
// short enough for the dead-predecessor sentinel, no connection-id counting -- as ONE block of gfx950 assembly (VBT_SWEEP_TEXT, the
// body of a single asm statement).  The C++ loop next to it in lattice.hip states the same recurrence over 64-byte scalar records
// and serves the other builds.  What the measurements behind this file say (s_memtime stamps around the loop's waits, round 4:
// DESIGN.md section 6): an iteration of the C++ loop spends a fifth to a quarter of its time parked on the scalar load of its pass
// record (an L2 round trip that nothing can be put in front of: scalar loads and LDS reads share one counter and return out of
// order with each other, so every LDS wait is a wait for the record in flight as well), and the instruction count of the common
// pass hardly matters next to that.  So here NOTHING comes through the scalar cache:
//
//   * a pass is a 16-byte record {w0, w1, meta, -} in global memory (VRec in lattice.hip: LDS address of the first predecessor's
//     slot record / of the first candidate's record; meta = candidates | predecessors of this round << 8 | (units | first round
//     << 3 | last round << 4) << 16 | phases that see a predecessor in the step << 24), fetched by ONE vector load that every lane
//     aims at the same address, four iterations before the pass is issued: vector loads return in order, so the wait for the
//     gathers of the pass in hand is the wait for the record as well -- and a long way behind both;
//   * lane masks are not data but two compares: lane (candidate cl, phase k) holds a pair in unit i iff cl < candidates and
//     4 i + k < predecessors (v_cmp on a byte of meta, SDWA); the mask of unit 0 is kept in an SGPR pair from issue to consume
//     (where it is the lanes that write, for the common pass);
//   * what is left to wait for per iteration: vmcnt at the top (loads issued two iterations ago), then the iteration's LDS reads.
//
// Software pipeline: gather depth VBT_DEPTH = 2, records fetched 4 ahead, iterations unrolled by four (gather slot = iteration
// & 1, record slot = record & 3).  Iteration si consumes pass si, issues the gathers of pass si + 2 from record si + 2 and
// requests record si + 4 -- the request in front of the gathers, so that whatever waits for the gathers of an iteration has
// waited for its record request too.
//
// Gathers: a NARROW pass (one unit) issues one load, a wide pass four (a unit without lanes runs under EXEC = 0: it moves nothing
// but takes its place in vmcnt, tools/calib/exec0_vmcnt.hip).  Loads return in order, so everything issued up to the gathers of
// pass si has landed once no more loads are in flight than iteration si - 1 issued: its record request + 1 or 4 gathers.  Which
// of the two is CONTROL FLOW, not data: every iteration exists twice, an N variant entered behind a narrow issue (s_waitcnt
// vmcnt(2)) and a W variant entered behind a wide one (vmcnt(5)); the out-of-line block that issues units 1..3 of a wide pass
// ends in a branch to the W variant of the next iteration.  So the wait is exact whatever the mix, and tools/check_ring_isa.py
// proves it on the compiled ISA by counting loads along paths.  The counter is drained behind the loop.
//
// The common pass -- at most 4 predecessors, at most 16 candidates, one step: 62 % of the passes of running text -- is a straight
// line without a taken branch; every other shape (more units, rounds of a step, empty passes behind the last) branches out of line.
//
// Registers (fixed, declared as clobbers; inputs are operands):
//   v24 zero   v25 cl   v26..v29 k, k + 4, k + 8, k + 12   v30 8 k   v31 8 cl
//   v32-35 / v36-39  gather slot 0 / 1: connection costs of units 0..3 of the pass in flight
//   v40 v41 v42 / v43 v44 v45  slot 0 / 1: this lane's predecessor address, candidate record address, meta of the pass in the slot
//   v46 candidate record address of the pass being issued   v47 first cell of its matrix row   v48-51 right ids / cell indices of its units
//   v52-59  slot records {field | right id, cost} of units 0..3 of the pass in hand     v60 its candidate's {slot offset | word cost << 16}
//   v61 v62 minimum cost / field of the winner   v63 v64 slot address, node cost   v65 scratch
//   v66:67  the lane's running minimum (field | right id, cost) across units and rounds of a step; all ones between steps
//   v[68:70] v[72:74] v[76:78] v[80:82]  record slots 0..3
//   s[36:37] / s[38:39]  slot 0 / 1: lanes of unit 0 of the pass in the slot     s40 / s41 its flags (units | first << 3 | last << 4)
//   s42-s54 scratch (meta, flags, masks)     s55 passes left     s[56:57] address of the record of the trip's first pass
#pragma once

#define VBT_DPP1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define VBT_DPP2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define VBT_SDWA_LO " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define VBT_SDWA_SEXT_HI " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define VBT_B0 " src0_sel:DWORD src1_sel:BYTE_0\n\t"
#define VBT_B1 " src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define VBT_B3 " src0_sel:DWORD src1_sel:BYTE_3\n\t"

// combine the four phases of every candidate and write its node: in KLO / KHI the lane's (field | right id, cost + connection cost),
// VM = lanes that saw a predecessor (and write).  Minimum cost over the quad, then the smallest field among the lanes that hold it
// (the last inserted predecessor, lattice.rs:141-146); + word cost (lattice.rs:125); cost -> the slot record, field -> the low half
// of the candidate record (the back pointer).  FILL1/FILL2: two independent instructions for the DPP wait states.
#define VBT_FINISH(KLO, KHI, VM, CA, FILL1, FILL2)                                                   \
    "v_cndmask_b32_e64 v65, -1, " KHI ", " VM "\n\t"                                                 \
    FILL1 FILL2                                                                                       \
    "v_min_u32_dpp v61, v65, v65" VBT_DPP1                                                            \
    "v_add_u32_sdwa v63, v60, %[offk]" VBT_SDWA_LO                                                    \
    "s_nop 0\n\t"                                                                                     \
    "v_min_u32_dpp v61, v61, v61" VBT_DPP2                                                            \
    "v_cmp_eq_u32_e32 vcc, v65, v61\n\t"                                                              \
    "v_add_u32_sdwa v64, v61, sext(v60)" VBT_SDWA_SEXT_HI                                             \
    "v_cndmask_b32_e32 v62, -1, " KLO ", vcc\n\t"                                                     \
    "s_nop 1\n\t"                                                                                     \
    "v_min_u32_dpp v62, v62, v62" VBT_DPP1                                                            \
    "s_nop 1\n\t"                                                                                     \
    "v_min_u32_dpp v62, v62, v62" VBT_DPP2                                                            \
    "s_mov_b64 exec, " VM "\n\t"                                                                      \
    "ds_write_b32 v63, v64 offset:4\n\t"                                                              \
    "ds_write_b16_d16_hi " CA ", v62\n\t"

// one unit of a general pass: (cost + connection cost, field) of predecessor 4 i + k against the running minimum; MASK = "" (a unit
// in front of the last: full for every candidate that exists -- lanes of the others compute garbage nobody writes) or the lanes
// that hold a pair in this, the last, unit: KREG = k + 4 i against the predecessors of the round, s[46:47] = cl < candidates
#define VBT_UNIT(KLO, KHI, W, MASK)                                                                  \
    "v_add_u32 v" KHI ", v" KHI ", " W "\n\t"                                                         \
    "v_cmp_lt_u64_e32 vcc, v[" KLO ":" KHI "], v[66:67]\n\t"                                          \
    MASK                                                                                              \
    "v_cndmask_b32_e32 v66, v66, v" KLO ", vcc\n\t"                                                   \
    "v_cndmask_b32_e32 v67, v67, v" KHI ", vcc\n\t"
#define VBT_LAST(KREG, META)                                                                         \
    "v_cmp_lt_u32_sdwa s[48:49], " KREG ", " META VBT_B1                                              \
    "s_and_b64 s[48:49], s[48:49], s[46:47]\n\t"
#define VBT_LASTMASK "s_and_b64 vcc, vcc, s[48:49]\n\ts_nop 0\n\t"

// the issue side of a pass up to the mask of its unit 0: addresses, left row, right id of predecessor k, flags -> FL, lanes of unit
// 0 -> s[44:45]; SCC = a wide pass (s43 != 0)
#define VBT_ISSUE_HEAD(PA, NCA, RW0, RW1, RMETA, FL)                                                 \
    "v_add_u32 " PA ", " RW0 ", v30\n\t"                                                              \
    "v_add_u32 " NCA ", " RW1 ", v31\n\t"                                                             \
    "ds_read_b32 v47, " NCA "\n\t"                                                                    \
    "ds_read_b32 v48, " PA "\n\t"                                                                     \
    "v_readfirstlane_b32 s42, " RMETA "\n\t"                                                          \
    "v_cmp_lt_u32_sdwa vcc, v25, " RMETA VBT_B0                                                       \
    "v_cmp_lt_u32_sdwa s[48:49], v26, " RMETA VBT_B1                                                  \
    "s_bfe_u32 " FL ", s42, 0x80010\n\t"                                                              \
    "s_and_b64 s[44:45], vcc, s[48:49]\n\t"                                                           \
    "s_and_b32 s43, " FL ", 6\n\t"
#define VBT_WIDE_READS(PA)                                                                           \
    "ds_read_b32 v49, " PA " offset:32\n\t"                                                           \
    "ds_read_b32 v50, " PA " offset:64\n\t"                                                           \
    "ds_read_b32 v51, " PA " offset:96\n\t"
// units 1..3 of a wide pass behind unit 0's gather: cell indices, lanes, gathers
#define VBT_WIDE(W1, W2, W3, RMETA)                                                                  \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_add_u32_sdwa v49, v49, v47" VBT_SDWA_LO                                                        \
    "v_add_u32_sdwa v50, v50, v47" VBT_SDWA_LO                                                        \
    "v_add_u32_sdwa v51, v51, v47" VBT_SDWA_LO                                                        \
    "v_cmp_lt_u32_sdwa s[46:47], v25, " RMETA VBT_B0                                                  \
    "v_cmp_lt_u32_sdwa s[48:49], v27, " RMETA VBT_B1                                                  \
    "v_cmp_lt_u32_sdwa s[50:51], v28, " RMETA VBT_B1                                                  \
    "v_cmp_lt_u32_sdwa s[52:53], v29, " RMETA VBT_B1                                                  \
    "s_and_b64 exec, s[46:47], s[48:49]\n\t"                                                          \
    "buffer_load_sshort " W1 ", v49, %[rs], 0 idxen\n\t"                                              \
    "s_and_b64 exec, s[46:47], s[50:51]\n\t"                                                          \
    "buffer_load_sshort " W2 ", v50, %[rs], 0 idxen\n\t"                                              \
    "s_and_b64 exec, s[46:47], s[52:53]\n\t"                                                          \
    "buffer_load_sshort " W3 ", v51, %[rs], 0 idxen\n\t"                                              \
    "s_mov_b64 exec, -1\n\t"

// One iteration, in line: the common pass.  U = iteration & 3, V = variant (N / W), VMC = loads the previous iteration issued,
// W0..W3 PA CA META M FL = the gather slot's registers, RW0 RW1 RMETA = the record of the pass to issue, RLOAD / OFF = slot and
// byte offset of the record to request, TAILN = what follows a narrow issue.
#define VBT_ITER(U, V, VMC, W0, W1, W2, W3, PA, CA, META, M, FL, RW0, RW1, RMETA, RLOAD, OFF, TAILN, TAILW)                 \
    "\n.LBBvbt_i" U V "_%=:\n\t"                                                                      \
    "s_waitcnt vmcnt(" VMC ")\n\t"                              /* the gathers of the pass in hand, the record of the pass to issue */ \
    "ds_read_b64 v[52:53], " PA "\n\t"                           /* predecessor k of the pass in hand */ \
    "ds_read_b32 v60, " CA " offset:4\n\t"                       /* its candidate: slot offset | word cost */ \
    "s_cmp_lg_u32 " FL ", 25\n\t"                                /* one unit that starts and ends the step? */ \
    "s_cbranch_scc1 .LBBvbt_g" U V "_%=\n\t"                                                          \
    VBT_ISSUE_HEAD(PA, "v46", RW0, RW1, RMETA, FL)                                                    \
    "s_cbranch_scc1 .LBBvbt_r" U V "_%=\n"                                                            \
    "\n.LBBvbt_b" U V "_%=:\n\t"                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    "global_load_dwordx3 " RLOAD ", v24, s[56:57] offset:" OFF "\n\t"   /* the record of four passes on */ \
    "v_add_u32 v53, v53, " W0 "\n\t"                             /* wrapping i32 add of the connection cost (lattice.rs:139) */ \
    VBT_FINISH("v52", "v53", M, CA, "v_add_u32_sdwa v48, v48, v47" VBT_SDWA_LO, "v_mov_b32 " META ", " RMETA "\n\t")        \
    "\n.LBBvbt_j" U V "_%=:\n\t"                                                                      \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                    \
    "buffer_load_sshort " W0 ", v48, %[rs], 0 idxen\n\t"                                              \
    "s_cmp_lg_u32 s43, 0\n\t"                                    /* more units: a wide pass */        \
    "s_cbranch_scc1 .LBBvbt_w" U V "_%=\n\t"                                                          \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_mov_b32 " CA ", v46\n\t"                                                                       \
    "s_mov_b64 " M ", s[44:45]\n\t"                                                                   \
    TAILN

// the out-of-line blocks of an iteration: the general consume side (any number of units, rounds of a step, empty passes), the right
// ids of a wide pass, its gathers
#define VBT_ITER_OOL(U, V, VMC, W0, W1, W2, W3, PA, CA, META, M, FL, RW0, RW1, RMETA, RLOAD, OFF, TAILN, TAILW)             \
    "\n.LBBvbt_g" U V "_%=:\n\t"                                                                      \
    "ds_read_b64 v[54:55], " PA " offset:32\n\t"                                                      \
    "ds_read_b64 v[56:57], " PA " offset:64\n\t"                                                      \
    "ds_read_b64 v[58:59], " PA " offset:96\n\t"                                                      \
    "s_mov_b32 s54, " FL "\n\t"                                  /* flags of the pass in hand */      \
    VBT_ISSUE_HEAD(PA, "v46", RW0, RW1, RMETA, FL)                                                    \
    "s_cbranch_scc0 .LBBvbt_n" U V "_%=\n\t"                                                          \
    VBT_WIDE_READS(PA)                                                                                \
    "\n.LBBvbt_n" U V "_%=:\n\t"                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    "global_load_dwordx3 " RLOAD ", v24, s[56:57] offset:" OFF "\n\t"                                 \
    "v_add_u32_sdwa v48, v48, v47" VBT_SDWA_LO                                                        \
    "v_cmp_lt_u32_sdwa s[46:47], v25, " META VBT_B0              /* candidates of the pass in hand */ \
    "s_and_b32 s42, s54, 7\n\t"                                  /* units (0: an empty pass -- unit 0 under no lanes) */ \
    "s_cmp_lt_u32 s42, 2\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l0" U V "_%=\n\t"                                                         \
    VBT_UNIT("52", "53", W0, "")                                                                      \
    "s_cmp_lt_u32 s42, 3\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l1" U V "_%=\n\t"                                                         \
    VBT_UNIT("54", "55", W1, "")                                                                      \
    "s_cmp_lt_u32 s42, 4\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l2" U V "_%=\n\t"                                                         \
    VBT_UNIT("56", "57", W2, "")                                                                      \
    VBT_LAST("v29", META)                                                                             \
    VBT_UNIT("58", "59", W3, VBT_LASTMASK)                                                            \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l0" U V "_%=:\n\t"                                                                     \
    VBT_LAST("v26", META)                                                                             \
    VBT_UNIT("52", "53", W0, VBT_LASTMASK)                                                            \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l1" U V "_%=:\n\t"                                                                     \
    VBT_LAST("v27", META)                                                                             \
    VBT_UNIT("54", "55", W1, VBT_LASTMASK)                                                            \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l2" U V "_%=:\n\t"                                                                     \
    VBT_LAST("v28", META)                                                                             \
    VBT_UNIT("56", "57", W2, VBT_LASTMASK)                                                            \
    "\n.LBBvbt_f" U V "_%=:\n\t"                                                                      \
    "s_bitcmp1_b32 s54, 4\n\t"                                   /* the last round of the step: combine, write, start over */ \
    "s_cbranch_scc0 .LBBvbt_m" U V "_%=\n\t"                                                          \
    "v_cmp_lt_u32_sdwa s[48:49], v26, " META VBT_B3              /* phases that saw a predecessor in the step */ \
    "s_and_b64 s[50:51], s[46:47], s[48:49]\n\t"                                                      \
    VBT_FINISH("v66", "v67", "s[50:51]", CA, "s_nop 0\n\t", "s_nop 0\n\t")                            \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_mov_b32 v66, -1\n\t"                                                                           \
    "v_mov_b32 v67, -1\n\t"                                                                           \
    "\n.LBBvbt_m" U V "_%=:\n\t"                                                                      \
    "v_mov_b32 " META ", " RMETA "\n\t"                                                               \
    "s_branch .LBBvbt_j" U V "_%=\n"                                                                  \
    "\n.LBBvbt_r" U V "_%=:\n\t"                                                                      \
    VBT_WIDE_READS(PA)                                                                                \
    "s_branch .LBBvbt_b" U V "_%=\n"                                                                  \
    "\n.LBBvbt_w" U V "_%=:\n\t"                                                                      \
    VBT_WIDE(W1, W2, W3, RMETA)                                                                       \
    "v_mov_b32 " CA ", v46\n\t"                                                                       \
    "s_mov_b64 " M ", s[44:45]\n\t"                                                                   \
    TAILW

// gather slots and record slots
#define VBT_G0 "v32", "v33", "v34", "v35", "v40", "v41", "v42", "s[36:37]", "s40"
#define VBT_G1 "v36", "v37", "v38", "v39", "v43", "v44", "v45", "s[38:39]", "s41"
#define VBT_R0 "v68", "v69", "v70"
#define VBT_R1 "v72", "v73", "v74"
#define VBT_R2 "v76", "v77", "v78"
#define VBT_R3 "v80", "v81", "v82"
// behind iterations 1 and 3: two passes done -- out if none are left; behind iteration 3 the records move on by four
#define VBT_HALF(NEXT)                                                                               \
    "s_sub_i32 s55, s55, 2\n\t"                                                                       \
    "s_cmp_gt_i32 s55, 0\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_i2" NEXT "_%=\n\t"                                                        \
    "s_branch .LBBvbt_x_%=\n"
#define VBT_HALF_FALL                                                                                \
    "s_sub_i32 s55, s55, 2\n\t"                                                                       \
    "s_cmp_gt_i32 s55, 0\n\t"                                                                         \
    "s_cbranch_scc0 .LBBvbt_x_%=\n\t"
#define VBT_TRIP(NEXT)                                                                               \
    "s_add_u32 s56, s56, 64\n\t"                                                                      \
    "s_addc_u32 s57, s57, 0\n\t"                                                                      \
    "s_sub_i32 s55, s55, 2\n\t"                                                                       \
    "s_cmp_gt_i32 s55, 0\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_i0" NEXT "_%=\n\t"                                                        \
    "s_branch .LBBvbt_x_%=\n"

#define VBT_EXPAND(M, ...) M(__VA_ARGS__)
// iteration U: gather slot U & 1, issues from record slot (U + 2) & 3, requests into record slot U
#define VBT_IT0(M, V, VMC, TN, TW) VBT_EXPAND(M, "0", V, VMC, VBT_G0, VBT_R2, "v[68:70]", "64", TN, TW)
#define VBT_IT1(M, V, VMC, TN, TW) VBT_EXPAND(M, "1", V, VMC, VBT_G1, VBT_R3, "v[72:74]", "80", TN, TW)
#define VBT_IT2(M, V, VMC, TN, TW) VBT_EXPAND(M, "2", V, VMC, VBT_G0, VBT_R0, "v[76:78]", "96", TN, TW)
#define VBT_IT3(M, V, VMC, TN, TW) VBT_EXPAND(M, "3", V, VMC, VBT_G1, VBT_R1, "v[80:82]", "112", TN, TW)

// the prologue's issue of pass P (record slot P, gather slot P): the record of pass P + 2 is requested first
#define VBT_PRO(P, W0, W1, W2, W3, PA, CA, META, M, FL, RW0, RW1, RMETA, RLOAD, OFF)                 \
    "global_load_dwordx3 " RLOAD ", v24, s[56:57] offset:" OFF "\n\t"                                 \
    VBT_ISSUE_HEAD(PA, CA, RW0, RW1, RMETA, FL)                                                       \
    "s_cbranch_scc0 .LBBvbt_pn" P "_%=\n\t"                                                           \
    VBT_WIDE_READS(PA)                                                                                \
    "\n.LBBvbt_pn" P "_%=:\n\t"                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    "v_add_u32_sdwa v48, v48, v47" VBT_SDWA_LO                                                        \
    "v_mov_b32 " META ", " RMETA "\n\t"                                                               \
    "s_mov_b64 " M ", s[44:45]\n\t"                                                                   \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                    \
    "buffer_load_sshort " W0 ", v48, %[rs], 0 idxen\n\t"                                              \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "s_cmp_lg_u32 s43, 0\n\t"
#define VBT_PRO0 VBT_EXPAND(VBT_PRO, "0", VBT_G0, VBT_R0, "v[76:78]", "32")
#define VBT_PRO1 VBT_EXPAND(VBT_PRO, "1", VBT_G1, VBT_R1, "v[80:82]", "48")

#define VBT_SWEEP_TEXT                                                                               \
    "v_mov_b32 v24, 0\n\t"                                                                            \
    "v_lshrrev_b32 v25, 2, %[ln]\n\t"                                                                 \
    "v_and_b32 v26, 3, %[ln]\n\t"                                                                     \
    "v_add_u32 v27, 4, v26\n\t"                                                                       \
    "v_add_u32 v28, 8, v26\n\t"                                                                       \
    "v_add_u32 v29, 12, v26\n\t"                                                                      \
    "v_lshlrev_b32 v30, 3, v26\n\t"                                                                   \
    "v_lshlrev_b32 v31, 3, v25\n\t"                                                                   \
    "v_mov_b32 v66, -1\n\t"                                                                           \
    "v_mov_b32 v67, -1\n\t"                                                                           \
    "s_mov_b64 s[56:57], %[rp]\n\t"                                                                   \
    "s_mov_b32 s55, %[sl]\n\t"                                                                        \
    "s_nop 0\n\t"                                                                                     \
    "global_load_dwordx3 v[68:70], v24, s[56:57]\n\t"                                                 \
    "global_load_dwordx3 v[72:74], v24, s[56:57] offset:16\n\t"                                       \
    "s_waitcnt vmcnt(0)\n\t"                                                                          \
    /* prologue: the gathers of passes 0 and 1 (behind the requests for the records of passes 2 and 3) */ \
    VBT_PRO0                                                                                          \
    "s_cbranch_scc0 .LBBvbt_p0_%=\n\t"                                                                \
    VBT_WIDE("v33", "v34", "v35", "v70")                                                              \
    "\n.LBBvbt_p0_%=:\n\t"                                                                            \
    VBT_PRO1                                                                                          \
    "s_cbranch_scc0 .LBBvbt_i0N_%=\n\t"                                                               \
    VBT_WIDE("v37", "v38", "v39", "v74")                                                              \
    "s_branch .LBBvbt_i0W_%=\n"                                                                       \
    /* the loop: the narrow variants in line */                                                       \
    VBT_IT0(VBT_ITER, "N", "2", "", "")                                                               \
    VBT_IT1(VBT_ITER, "N", "2", VBT_HALF_FALL, "")                                                    \
    VBT_IT2(VBT_ITER, "N", "2", "", "")                                                               \
    VBT_IT3(VBT_ITER, "N", "2", VBT_TRIP("N"), "")                                                    \
    VBT_IT0(VBT_ITER, "W", "5", "s_branch .LBBvbt_i1N_%=\n", "")                                      \
    VBT_IT1(VBT_ITER, "W", "5", VBT_HALF("N"), "")                                                    \
    VBT_IT2(VBT_ITER, "W", "5", "s_branch .LBBvbt_i3N_%=\n", "")                                      \
    VBT_IT3(VBT_ITER, "W", "5", VBT_TRIP("N"), "")                                                    \
    VBT_IT0(VBT_ITER_OOL, "N", "2", "", "s_branch .LBBvbt_i1W_%=\n")                                  \
    VBT_IT1(VBT_ITER_OOL, "N", "2", "", VBT_HALF("W"))                                                \
    VBT_IT2(VBT_ITER_OOL, "N", "2", "", "s_branch .LBBvbt_i3W_%=\n")                                  \
    VBT_IT3(VBT_ITER_OOL, "N", "2", "", VBT_TRIP("W"))                                                \
    VBT_IT0(VBT_ITER_OOL, "W", "5", "", "s_branch .LBBvbt_i1W_%=\n")                                  \
    VBT_IT1(VBT_ITER_OOL, "W", "5", "", VBT_HALF("W"))                                                \
    VBT_IT2(VBT_ITER_OOL, "W", "5", "", "s_branch .LBBvbt_i3W_%=\n")                                  \
    VBT_IT3(VBT_ITER_OOL, "W", "5", "", VBT_TRIP("W"))                                                \
    "\n.LBBvbt_x_%=:\n\t"                                                                             \
    "s_waitcnt vmcnt(0)\n\t"                                                                          \
    "s_mov_b64 exec, -1"

#define VBT_SWEEP_CLOBBERS                                                                           \
    "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",                \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",   \
    "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v72", "v73", "v74",                 \
    "v76", "v77", "v78", "v80", "v81", "v82",                                                                                       \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",               \
    "s52", "s53", "s54", "s55", "s56", "s57", "vcc", "scc", "memory"
