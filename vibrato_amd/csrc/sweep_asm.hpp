// The sweep loop of lattice_sentence (lattice.hip) for the common build -- i16 connection cells, sentences
// short enough for the dead-predecessor sentinel -- as ONE block of gfx950 assembly (VBT_SWEEP_TEXT, the body of a single asm
// statement).  The C++ loop next to it in lattice.hip states the same recurrence and serves the other builds; this one exists
// because the loop is bound by instruction issue (20 waves of a CU share one scalar unit and four vector ALUs: what a pass costs
// is what it issues), and what the compiler makes of the C++ is ~60 instructions and two taken branches for the pass that running
// text is made of: at most 4 predecessors, at most 16 candidates, one step.  Here that pass is a straight line of ~37 instructions
// with no taken branch; every other shape branches out of line.
//
// Software pipeline: gather depth VBT_DEPTH = 2, iterations unrolled by two.  Iteration si consumes pass si and issues the gathers
// of pass si + 2; the record in hand is record si + 2 (its issue half belongs to pass si + 2, its consume half to pass si: LPass in
// device_common.hpp), in one of two 16-SGPR buffers; the next record is requested into the other buffer once every LDS read of
// the iteration is back (scalar loads and LDS reads share lgkmcnt and return out of order with each other).  (Requesting it at
// the top of the iteration instead, a whole iteration ahead, and waiting for the LDS reads with lgkmcnt(1) behind one extra read --
// sound, since LDS reads return in order -- measured 6 % SLOWER: the reads come back behind the scalar load.)
//
// Gathers: a NARROW pass (one unit) issues one load, a wide pass four (units without lanes run under EXEC = 0: they move nothing
// but take their place in vmcnt, tools/calib/exec0_vmcnt.hip).  Loads return in order, so the gathers of pass si have landed
// once no more loads are in flight than pass si + 1 issued -- 1 or 4.  Which of the two is CONTROL FLOW, not data: every
// iteration exists twice, an N variant entered behind a narrow issue (s_waitcnt vmcnt(1)) and a W variant entered behind a wide
// one (vmcnt(4)); the out-of-line block that issues units 1..3 of a wide pass ends in a branch to the W variant of the next
// iteration.  So the wait is exact whatever the mix, and tools/check_ring_isa.py can prove it on the compiled ISA by counting
// loads along paths.  The counter is drained behind the loop.
//
// Registers (fixed, declared as clobbers; inputs are operands):
//   v40-43 / v44-47  ring slot 0 / 1: connection costs of units 0..3 of the pass in flight
//   v48, v49 / v50, v51  slot 0 / 1: this lane's predecessor address (slot record of predecessor k) and candidate record address
//   v52  candidate record address of the pass being issued      v53  first cell of its matrix row
//   v54-57  right ids (then cell indices) of its units 0..3      v58-65  slot records {field | right id, cost} of units 0..3 of the pass in hand
//   v66  the candidate's {slot offset | word cost << 16}          v67, v68  minimum cost / field of the winner     v69, v70  slot address, node cost
//   v72:73  the lane's running minimum (field | right id, cost) across units and rounds of a step; all ones between steps
//   s[36:51] / s[52:67]  record buffers A / B (w0 w1 w0c w1c m0 m1 m2 m3 lm vm)     s68  flags / units     s69  passes left
//   s[74:75]  address of the record of the trip's first pass
#pragma once

#define VBT_DPP1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define VBT_DPP2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define VBT_SDWA_LO " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define VBT_SDWA_SEXT_HI " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"

// developer aid (VBT_LOOP_PROF builds, tools/phase_profile.py): cycles an iteration is parked at its two waits, summed in
// s[80:81] (the record) and s[82:83] (the gathers + the LDS reads); s_memtime stamps the moment it issues
#if VBT_LOOP_PROF
#define VBT_PROF_WAIT(ACC_LO, ACC_HI, WAIT)                                                          \
    "s_memtime s[76:77]\n\t" WAIT "s_memtime s[78:79]\n\ts_waitcnt lgkmcnt(0)\n\t"                   \
    "s_sub_u32 s76, s78, s76\n\ts_subb_u32 s77, s79, s77\n\t"                                        \
    "s_add_u32 " ACC_LO ", " ACC_LO ", s76\n\ts_addc_u32 " ACC_HI ", " ACC_HI ", s77\n\t"
#define VBT_PROF_INIT "s_mov_b64 s[80:81], 0\n\ts_mov_b64 s[82:83], 0\n\ts_mov_b64 s[84:85], 0\n\ts_mov_b64 s[86:87], 0\n\t"      \
                      "s_mov_b64 s[88:89], 0\n\ts_mov_b64 s[90:91], 0\n\ts_mov_b64 s[92:93], 0\n\t"
// (the common pass in two more pieces: s[84:85] = landed wait .. node writes (the VALU chain), s[86:87] = node writes .. end of the
// iteration (writes, gather, bookkeeping); stamps of an iteration are summed behind the next one's record wait)
#define VBT_PROF_STAMP(R) "s_memtime " R "\n\t"
#if VBT_LOOP_PROF == 1  // which pass is cut in two: the common one (1) or the general one (2)
#define VBT_PROF_STAMP_C(R) VBT_PROF_STAMP(R)
#define VBT_PROF_STAMP_G(R)
#else
#define VBT_PROF_STAMP_C(R)
#define VBT_PROF_STAMP_G(R) VBT_PROF_STAMP(R)
#endif
#define VBT_PROF_SUM                                                                                 \
    "s_sub_u32 s76, s90, s88\n\ts_subb_u32 s77, s91, s89\n\ts_add_u32 s84, s84, s76\n\ts_addc_u32 s85, s85, s77\n\t"  \
    "s_sub_u32 s76, s92, s90\n\ts_subb_u32 s77, s93, s91\n\ts_add_u32 s86, s86, s76\n\ts_addc_u32 s87, s87, s77\n\t"  \
    "s_mov_b64 s[88:89], 0\n\ts_mov_b64 s[90:91], 0\n\ts_mov_b64 s[92:93], 0\n\t"
#define VBT_PROF_OUT "v_mov_b32 v58, s80\n\tv_mov_b32 v59, s81\n\tv_mov_b32 v60, s82\n\tv_mov_b32 v61, s83\n\t"          \
                     "v_mov_b32 v62, s84\n\tv_mov_b32 v63, s85\n\tv_mov_b32 v64, s86\n\tv_mov_b32 v65, s87\n\t"          \
                     "ds_write_b64 %[plds], v[62:63] offset:16\n\tds_write_b64 %[plds], v[64:65] offset:24\n\t"          \
                     "ds_write_b64 %[plds], v[58:59]\n\tds_write_b64 %[plds], v[60:61] offset:8\n\ts_waitcnt lgkmcnt(0)\n\t"
#define VBT_PROF_CLOBBERS "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93",
#else
#define VBT_PROF_WAIT(ACC_LO, ACC_HI, WAIT) WAIT
#define VBT_PROF_INIT
#define VBT_PROF_STAMP(R)
#define VBT_PROF_STAMP_C(R)
#define VBT_PROF_STAMP_G(R)
#define VBT_PROF_SUM
#define VBT_PROF_OUT
#define VBT_PROF_CLOBBERS
#endif

// combine the four phases of every candidate and write its node: in v[KLO:KHI] the lane's (field | right id, cost + connection cost),
// VM = lanes that saw a predecessor (and write).  Minimum cost over the quad, then the smallest field among the lanes that hold it
// (the last inserted predecessor, lattice.rs:141-146); + word cost (lattice.rs:125); cost -> the slot record, field -> the low half
// of the candidate record (the back pointer).  FILL1/FILL2: two independent instructions for the DPP wait states.
#define VBT_FINISH(KLO, KHI, VM, CA, FILL1, FILL2, STAMP)                                                  \
    "v_cndmask_b32_e64 v59, -1, " KHI ", " VM "\n\t"                                                 \
    FILL1 FILL2                                                                                       \
    "v_min_u32_dpp v67, v59, v59" VBT_DPP1                                                            \
    "v_add_u32_sdwa v69, v66, %[offk]" VBT_SDWA_LO                                                    \
    "s_nop 0\n\t"                                                                                     \
    "v_min_u32_dpp v67, v67, v67" VBT_DPP2                                                            \
    "v_cmp_eq_u32_e32 vcc, v59, v67\n\t"                                                              \
    "v_add_u32_sdwa v70, v67, sext(v66)" VBT_SDWA_SEXT_HI                                             \
    "v_cndmask_b32_e32 v68, -1, " KLO ", vcc\n\t"                                                     \
    "s_nop 1\n\t"                                                                                     \
    "v_min_u32_dpp v68, v68, v68" VBT_DPP1                                                            \
    "s_nop 1\n\t"                                                                                     \
    "v_min_u32_dpp v68, v68, v68" VBT_DPP2                                                            \
    STAMP                                                                                             \
    "s_mov_b64 exec, " VM "\n\t"                                                                      \
    "ds_write_b32 v69, v70 offset:4\n\t"                                                              \
    "ds_write_b16_d16_hi " CA ", v68\n\t"

// one unit of a general pass: (cost + connection cost, field) of predecessor 4 i + k against the running minimum; MASK = "" or an
// s_and of vcc with the lanes that hold a pair in this (last) unit
#define VBT_UNIT(KLO, KHI, W, MASK)                                                                  \
    "v_add_u32 v" KHI ", v" KHI ", " W "\n\t"                                                         \
    "v_cmp_lt_u64_e32 vcc, v[" KLO ":" KHI "], v[72:73]\n\t"                                          \
    MASK                                                                                              \
    "v_cndmask_b32_e32 v72, v72, v" KLO ", vcc\n\t"                                                   \
    "v_cndmask_b32_e32 v73, v73, v" KHI ", vcc\n\t"

// units 1..3 of a wide pass.  Their right ids are read at the top of the iteration with every other LDS read (VBT_WIDE_READS, out of
// line: a narrow pass skips them); behind unit 0's gather the cell indices and the three gathers (EXEC = the lanes of each unit).
#define VBT_WIDE_READS(PA)                                                                           \
    "ds_read_b32 v55, " PA " offset:32\n\t"                                                           \
    "ds_read_b32 v56, " PA " offset:64\n\t"                                                           \
    "ds_read_b32 v57, " PA " offset:96\n\t"
#define VBT_WIDE(PA, W1, W2, W3, M1, M2, M3)                                                         \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_add_u32_sdwa v55, v55, v53" VBT_SDWA_LO                                                        \
    "v_add_u32_sdwa v56, v56, v53" VBT_SDWA_LO                                                        \
    "v_add_u32_sdwa v57, v57, v53" VBT_SDWA_LO                                                        \
    "s_mov_b64 exec, " M1 "\n\t"                                                                      \
    "buffer_load_sshort " W1 ", v55, %[rs], 0 idxen\n\t"                                              \
    "s_mov_b64 exec, " M2 "\n\t"                                                                      \
    "buffer_load_sshort " W2 ", v56, %[rs], 0 idxen\n\t"                                              \
    "s_mov_b64 exec, " M3 "\n\t"                                                                      \
    "buffer_load_sshort " W3 ", v57, %[rs], 0 idxen\n\t"                                              \
    "s_mov_b64 exec, -1\n\t"

// the issue side of a pass up to its first gather: addresses, left row, right id of predecessor k (used by the prologue; the
// iterations interleave the same instructions with the consume side)
#define VBT_ISSUE0(PA, CA, RW0, RW1, RM0, W0)                                                        \
    "v_add_u32 " PA ", " RW0 ", %[k8]\n\t"                                                            \
    "v_add_u32 " CA ", " RW1 ", %[cl8]\n\t"                                                           \
    "ds_read_b32 v53, " CA "\n\t"                                                                     \
    "ds_read_b32 v54, " PA "\n\t"                                                                     \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    "v_add_u32_sdwa v54, v54, v53" VBT_SDWA_LO                                                        \
    "s_mov_b64 exec, " RM0 "\n\t"                                                                     \
    "buffer_load_sshort " W0 ", v54, %[rs], 0 idxen\n\t"

// One iteration.  U = slot (0, 1), V = variant (N: vmcnt 1, W: vmcnt 4), W0..W3 / PA / CA = the slot's ring registers, R* = the
// fields of the record in hand, NX / OFF = the buffer and byte offset of the next record, TAILN / TAILW = what follows a narrow /
// a wide issue (ring bookkeeping + where to go).
#define VBT_ITER(U, V, VMC, W0, W1, W2, W3, PA, CA, RW0, RW1, RFL, RM0, RM1, RM2, RM3, RLM, RVM, NX, OFF, TAILN, TAILW)      \
    "\n.LBBvbt_i" U V "_%=:\n\t"                                                                        \
    VBT_PROF_WAIT("s80", "s81", "s_waitcnt lgkmcnt(0)\n\t")      /* the record in hand has arrived */ \
    VBT_PROF_SUM                                                                                      \
    "ds_read_b64 v[58:59], " PA "\n\t"                           /* predecessor k of the pass in hand */ \
    "ds_read_b32 v66, " CA " offset:4\n\t"                       /* its candidate: slot offset | word cost */ \
    "s_lshr_b32 s68, " RFL ", 20\n\t"                                                                 \
    "s_cmp_lg_u32 s68, 25\n\t"                                   /* one unit that starts and ends the step? */ \
    "s_cbranch_scc1 .LBBvbt_g" U V "_%=\n\t"                                                          \
    "v_add_u32 " PA ", " RW0 ", %[k8]\n\t"                       /* the pass to issue: addresses, left row, right id */ \
    "v_add_u32 v52, " RW1 ", %[cl8]\n\t"                                                              \
    "ds_read_b32 v53, v52\n\t"                                                                        \
    "ds_read_b32 v54, " PA "\n\t"                                                                     \
    "s_cmp_lg_u64 " RM1 ", 0\n\t"                                /* a wide pass: the right ids of units 1..3 too */ \
    "s_cbranch_scc1 .LBBvbt_r" U V "_%=\n"                                                            \
    ".LBBvbt_b" U V "_%=:\n\t"                                                                        \
    VBT_PROF_WAIT("s82", "s83", "s_waitcnt vmcnt(" VMC ") lgkmcnt(0)\n\t")  /* the gathers of the pass in hand; every LDS read */ \
    VBT_PROF_STAMP_C("s[88:89]")                                                                        \
    "s_load_dwordx16 " NX ", s[74:75], " OFF "\n\t"                                                   \
    "v_add_u32 v59, v59, " W0 "\n\t"                             /* wrapping i32 add of the connection cost (lattice.rs:139) */ \
    VBT_FINISH("v58", "v59", RVM, CA,                                                                \
               "v_add_u32_sdwa v54, v54, v53" VBT_SDWA_LO, "s_nop 0\n\t", VBT_PROF_STAMP_C("s[90:91]")) \
    "\n.LBBvbt_j" U V "_%=:\n\t"                                                                        \
    "s_mov_b64 exec, " RM0 "\n\t"                                                                     \
    "buffer_load_sshort " W0 ", v54, %[rs], 0 idxen\n\t"                                              \
    "s_cmp_lg_u64 " RM1 ", 0\n\t"                                /* a second unit: a wide pass */     \
    "s_cbranch_scc1 .LBBvbt_w" U V "_%=\n\t"                                                          \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_mov_b32 " CA ", v52\n\t"                                                                       \
    VBT_PROF_STAMP("s[92:93]")                                                                        \
    TAILN

// the out-of-line blocks of an iteration: the general consume side (any number of units, rounds of a step) and the wide issue
#define VBT_ITER_OOL(U, V, VMC, W0, W1, W2, W3, PA, CA, RW0, RW1, RFL, RM0, RM1, RM2, RM3, RLM, RVM, NX, OFF, TAILN, TAILW)  \
    "\n.LBBvbt_g" U V "_%=:\n\t"                                                                        \
    "ds_read_b64 v[60:61], " PA " offset:32\n\t"                                                      \
    "ds_read_b64 v[62:63], " PA " offset:64\n\t"                                                      \
    "ds_read_b64 v[64:65], " PA " offset:96\n\t"                                                      \
    "v_add_u32 " PA ", " RW0 ", %[k8]\n\t"                                                            \
    "v_add_u32 v52, " RW1 ", %[cl8]\n\t"                                                              \
    "ds_read_b32 v53, v52\n\t"                                                                        \
    "ds_read_b32 v54, " PA "\n\t"                                                                     \
    "s_cmp_lg_u64 " RM1 ", 0\n\t"                                                                     \
    "s_cbranch_scc0 .LBBvbt_n" U V "_%=\n\t"                                                          \
    VBT_WIDE_READS(PA)                                                                                \
    "\n.LBBvbt_n" U V "_%=:\n\t"                                                                      \
    VBT_PROF_WAIT("s82", "s83", "s_waitcnt vmcnt(" VMC ") lgkmcnt(0)\n\t")                            \
    VBT_PROF_STAMP_G("s[88:89]")                                                                      \
    "s_load_dwordx16 " NX ", s[74:75], " OFF "\n\t"                                                   \
    "v_add_u32_sdwa v54, v54, v53" VBT_SDWA_LO                                                        \
    "s_and_b32 s68, s68, 7\n\t"                                  /* units (0: an empty pass -- unit 0 under no lanes) */ \
    "s_cmp_lt_u32 s68, 2\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l0" U V "_%=\n\t"                                                         \
    VBT_UNIT("58", "59", W0, "")                                                                      \
    "s_cmp_lt_u32 s68, 3\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l1" U V "_%=\n\t"                                                         \
    VBT_UNIT("60", "61", W1, "")                                                                      \
    "s_cmp_lt_u32 s68, 4\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l2" U V "_%=\n\t"                                                         \
    VBT_UNIT("62", "63", W2, "")                                                                      \
    VBT_UNIT("64", "65", W3, "s_and_b64 vcc, vcc, " RLM "\n\ts_nop 0\n\t")                                       \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l0" U V "_%=:\n\t"                                                                       \
    VBT_UNIT("58", "59", W0, "s_and_b64 vcc, vcc, " RLM "\n\ts_nop 0\n\t")                                       \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l1" U V "_%=:\n\t"                                                                       \
    VBT_UNIT("60", "61", W1, "s_and_b64 vcc, vcc, " RLM "\n\ts_nop 0\n\t")                                       \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l2" U V "_%=:\n\t"                                                                       \
    VBT_UNIT("62", "63", W2, "s_and_b64 vcc, vcc, " RLM "\n\ts_nop 0\n\t")                                       \
    "\n.LBBvbt_f" U V "_%=:\n\t"                                                                        \
    VBT_PROF_STAMP_G("s[90:91]")                                                                      \
    "s_bitcmp1_b32 " RFL ", 24\n\t"                              /* the last round of the step: combine, write, start over */ \
    "s_cbranch_scc0 .LBBvbt_j" U V "_%=\n\t"                                                          \
    VBT_FINISH("v72", "v73", RVM, CA, "s_nop 0\n\t", "s_nop 0\n\t", "")                               \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_mov_b32 v72, -1\n\t"                                                                           \
    "v_mov_b32 v73, -1\n\t"                                                                           \
    "s_branch .LBBvbt_j" U V "_%=\n"                                                                  \
    "\n.LBBvbt_r" U V "_%=:\n\t"                                                                        \
    VBT_WIDE_READS(PA)                                                                                \
    "s_branch .LBBvbt_b" U V "_%=\n"                                                                  \
    "\n.LBBvbt_w" U V "_%=:\n\t"                                                                        \
    VBT_WIDE(PA, W1, W2, W3, RM1, RM2, RM3)                                                           \
    "v_mov_b32 " CA ", v52\n\t"                                                                       \
    VBT_PROF_STAMP("s[92:93]")                                                                        \
    TAILW

// record buffers
#define VBT_RA "s36", "s37", "s39", "s[40:41]", "s[42:43]", "s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]"
#define VBT_RB "s52", "s53", "s55", "s[56:57]", "s[58:59]", "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]"
#define VBT_S0 "v40", "v41", "v42", "v43", "v48", "v49"
#define VBT_S1 "v44", "v45", "v46", "v47", "v50", "v51"
// the end of a trip (behind slot 1): two passes further; again while passes are left
#define VBT_TRIP(NEXT)                                                                               \
    "s_add_u32 s74, s74, 128\n\t"                                                                     \
    "s_addc_u32 s75, s75, 0\n\t"                                                                      \
    "s_sub_i32 s69, s69, 2\n\t"                                                                       \
    "s_cmp_gt_i32 s69, 0\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_i0" NEXT "_%=\n\t"                                                        \
    "s_branch .LBBvbt_x_%=\n"

#define VBT_EXPAND(M, ...) M(__VA_ARGS__)
#define VBT_IT0(M, V, VMC, TN, TW) VBT_EXPAND(M, "0", V, VMC, VBT_S0, VBT_RA, "s[52:67]", "0xc0", TN, TW)
#define VBT_IT1(M, V, VMC, TN, TW) VBT_EXPAND(M, "1", V, VMC, VBT_S1, VBT_RB, "s[36:51]", "0x100", TN, TW)

#define VBT_SWEEP_TEXT                                                                               \
    "s_mov_b64 s[74:75], %[rp]\n\t"                                                                   \
    VBT_PROF_INIT                                                                                     \
    "s_mov_b32 s69, %[sl]\n\t"                                                                        \
    "s_load_dwordx16 s[36:51], s[74:75], 0x0\n\t"                                                     \
    "s_load_dwordx16 s[52:67], s[74:75], 0x40\n\t"                                                    \
    "v_mov_b32 v72, -1\n\t"                                                                           \
    "v_mov_b32 v73, -1\n\t"                                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    /* prologue: the gathers of passes 0 and 1 */                                                     \
    VBT_ISSUE0("v48", "v49", "s36", "s37", "s[40:41]", "v40")                                         \
    "s_cmp_lg_u64 s[42:43], 0\n\t"                                                                    \
    "s_cbranch_scc0 .LBBvbt_p0_%=\n\t"                                                                \
    "s_mov_b64 exec, -1\n\t" VBT_WIDE_READS("v48") "s_waitcnt lgkmcnt(0)\n\t"                          \
    VBT_WIDE("v48", "v41", "v42", "v43", "s[42:43]", "s[44:45]", "s[46:47]")                          \
    "\n.LBBvbt_p0_%=:\n\t"                                                                              \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "s_load_dwordx16 s[36:51], s[74:75], 0x80\n\t"                                                    \
    VBT_ISSUE0("v50", "v51", "s52", "s53", "s[56:57]", "v44")                                         \
    "s_cmp_lg_u64 s[58:59], 0\n\t"                                                                    \
    "s_cbranch_scc0 .LBBvbt_p1_%=\n\t"                                                                \
    "s_mov_b64 exec, -1\n\t" VBT_WIDE_READS("v50") "s_waitcnt lgkmcnt(0)\n\t"                          \
    VBT_WIDE("v50", "v45", "v46", "v47", "s[58:59]", "s[60:61]", "s[62:63]")                          \
    "s_branch .LBBvbt_i0W_%=\n"                                                                       \
    "\n.LBBvbt_p1_%=:\n\t"                                                                              \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    /* the loop: the narrow variants in line */                                                       \
    VBT_IT0(VBT_ITER, "N", "1", "", "")                                                               \
    VBT_IT1(VBT_ITER, "N", "1", VBT_TRIP("N"), "")                                                    \
    VBT_IT0(VBT_ITER, "W", "4", "s_branch .LBBvbt_i1N_%=\n", "")                                      \
    VBT_IT1(VBT_ITER, "W", "4", VBT_TRIP("N"), "")                                                    \
    VBT_IT0(VBT_ITER_OOL, "N", "1", "", "s_branch .LBBvbt_i1W_%=\n")                                  \
    VBT_IT1(VBT_ITER_OOL, "N", "1", "", VBT_TRIP("W"))                                                \
    VBT_IT0(VBT_ITER_OOL, "W", "4", "", "s_branch .LBBvbt_i1W_%=\n")                                  \
    VBT_IT1(VBT_ITER_OOL, "W", "4", "", VBT_TRIP("W"))                                                \
    "\n.LBBvbt_x_%=:\n\t"                                                                               \
    "s_waitcnt vmcnt(0)\n\t"                                                                          \
    VBT_PROF_OUT                                                                                      \
    "s_mov_b64 exec, -1"

#define VBT_SWEEP_CLOBBERS                                                                           \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",   \
    "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v72", "v73",                       \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",               \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",               \
    "s68", "s69", "s74", "s75", VBT_PROF_CLOBBERS "vcc", "scc", "memory"
