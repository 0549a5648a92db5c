// The sweep loop of lattice_sentence (lattice.hip) for the common build -- i16 connection cells, sentences
// short enough for the dead-predecessor sentinel, an LDS tier of at most 64 KiB, no connection-id counting -- as ONE block of gfx950
// assembly (VBT_SWEEP_TEXT, the body of a single asm statement).  The C++ loop next to it in lattice.hip states the same recurrence
// over 64-byte scalar records and serves the other builds.  What the measurements behind this file say (s_memtime stamps around the
// loop's waits, round 4: DESIGN.md section 6): an iteration of the C++ loop spends a fifth to a quarter of its time parked on the
// scalar load of its pass record (an L2 round trip that nothing can be put in front of: scalar loads and LDS reads share one
// counter and return out of order with each other, so every LDS wait is a wait for the record in flight as well), and the
// instruction count of the common pass hardly matters next to that.  So here NOTHING comes through the scalar cache:
//
//   * a pass is an 8-byte record in global memory (VRec in lattice.hip): {w0 | phases that see a predecessor in the step << 16 |
//     candidates << 24,  w1 | (units | first round << 3 | last round << 4) << 16 | predecessors of this round << 24}, w0 / w1 = LDS
//     address of the first predecessor's slot record / of the first candidate's record (16 bits: the tier's LDS is at most 64 KiB).
//     ONE vector load that every lane aims at the same address fetches it, six iterations before the pass is issued: vector loads
//     return in order, so the wait for the gathers of the pass in hand is the wait for the record as well -- and a long way behind both;
//   * lane masks are not data but two compares: lane (candidate cl, phase k) holds a pair in unit i iff cl < candidates and
//     4 i + k < predecessors (v_cmp on a byte of the record, SDWA); the mask of unit 0 is kept in an SGPR pair from issue to
//     consume (where it is the lanes that write, for the common pass);
//   * what is left to wait for per iteration: vmcnt at the top (loads issued three iterations ago), then the iteration's LDS reads.
//
// Software pipeline: the gathers of a pass are issued 3 iterations before it is consumed (part of them miss L2: two iterations of
// lead left ~165 cycles per iteration parked at the top); iterations are unrolled by six (gather slot = iteration mod 3).  Iteration si
// consumes pass si and issues the gathers of pass si + 3 from record si + 3.  The records: with VBT_LDS_REC (the default build: 8-byte
// records in the sentence's LDS) record r sits in register slot r mod 3 and iteration si requests record si + 5 -- a broadcast ds_read
// behind the iteration's own LDS wait, landed at the next iteration's, first touched two iterations on; with the records in global
// memory (VBT_LDS_REC=0, round 4) the request is a vector load six passes ahead into one of six slots, in front of the gathers, so
// that whatever waits for the gathers of an iteration has waited for its record request too.
//
// Gathers: a NARROW pass (one unit) issues one load, a wide pass four (a unit without lanes runs under EXEC = 0: it moves nothing
// but takes its place in vmcnt, tools/calib/exec0_vmcnt.hip).  Loads return in order, so everything issued up to the gathers of
// pass si has landed once no more loads are in flight than iterations si - 2 and si - 1 issued: per iteration a record request +
// 1 or 4 gathers.  The width of iteration si - 1 is CONTROL FLOW, not data: every iteration exists twice, an N variant entered
// behind a narrow issue and a W variant entered behind a wide one; the out-of-line block that issues units 1..3 of a wide pass
// ends in a branch to the W variant of the next iteration.  Iteration si - 2 is counted as narrow (2 loads) whatever it was: if it
// was wide the wait asks for three of its gathers -- an iteration older than it has to -- a little early, never late.  So
// s_waitcnt vmcnt(4) / vmcnt(7), and tools/check_ring_isa.py proves on the compiled ISA, by counting loads along paths, that no ring
// register is touched before its load has landed.  The counter is drained behind the loop.
//
// The common pass -- at most 4 predecessors, at most 16 candidates, one step: 62 % of the passes of running text -- is a straight
// line without a taken branch; every other shape (more units, rounds of a step, empty passes behind the last) branches out of line.
//
// Registers (fixed, declared as clobbers; inputs are operands).  Default build (records in LDS): v24-v79 -- with the compiler's own
// that is 80 VGPRs, six waves per SIMD for the lean instance (round 5's layout, kept for VBT_LDS_REC=0, ran to v87):
//   v24 node cost (VBT_LDS_REC=0: zero)   v25 cl   v26..v29 k, k + 4, k + 8, k + 12   v30 8 k   v31 8 cl
//   v32-35 / v36-39 / v40-43  gather slot 0 / 1 / 2: connection costs of units 0..3 of the pass in flight
//   v44 v45 v46 / v47 v48 v49 / v50 v51 v52  slot 0 / 1 / 2: this lane's predecessor address, candidate record address, meta of the
//            pass in the slot (predecessors of the round << 8 | phases << 16 | candidates << 24)
//   v53 candidate record address of the pass being issued   v54 first cell of its matrix row   v55-58 right ids / cell indices of its units
//   v60-67  slot records {field | right id, cost} of units 0..3 of the pass in hand     v68 its candidate's {slot offset | word cost << 16}
//   v69 v70 minimum cost / field of the winner (v70 first holds the masked cost the minimum is taken of)   v71 slot address
//   v74:75  the lane's running minimum (field | right id, cost) across units and rounds of a step; all ones between steps
//   v[76:77] v[78:79] v[72:73]  record slots 0..2        (VBT_LDS_REC=0: v72 node cost, v73 scratch, v[76:77] .. v[86:87] record slots 0..5)
//   s[36:37] / s[38:39] / s[40:41]  slot 0 / 1 / 2: lanes of unit 0 of the pass in the slot     s42 / s43 / s44 its flags
//   s45-s57 scratch (record word, flags, masks)   s58 byte selector of meta   s59 passes left
//   v59 LDS address of the record of the trip's first pass (VBT_LDS_REC=0: s[60:61], its address in global memory)
#pragma once

#define VBT_DPP1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define VBT_DPP2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define VBT_SDWA_LO " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define VBT_SDWA_SEXT_HI " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
// -DVBT_NO_GATHER=1 (ceiling experiment, WRONG RESULTS by design: what any staging of the connection matrix could buy at most): every
// connection cost is 0, no gather is issued; an iteration then issues one load (its record request), so both variants wait with vmcnt(2)
// -DVBT_LDS_REC=1 (the default build): the pass records of a (segment of a) sentence sit in its LDS, behind the lattice arrays --
// 8 bytes per pass, ~0.5 KiB for the mean sentence -- instead of its region of global memory: the builder's stores are ds_writes,
// nothing has to be drained before the loop, the fetch of a record is a broadcast ds_read_b64 (v59 = LDS address of the first record
// of the trip) that the iteration's own LDS wait covers long before the record is used, and the gathers are the only vector loads
// left: an iteration issues 1 or 4, so the waits at the top are vmcnt(2) / vmcnt(5).  VBT_LDS_REC=0: round 4's records in global
// memory (a global_load_dwordx2 that every lane aims at one address; one load more per iteration in vmcnt).
#if VBT_LDS_REC
#define VBT_RECLOAD(RLOAD, OFF) "ds_read_b64 " RLOAD ", v59 offset:" OFF "\n\t"
#define VBT_REC_LOADS 0
#else
#define VBT_RECLOAD(RLOAD, OFF) "global_load_dwordx2 " RLOAD ", v24, s[60:61] offset:" OFF "\n\t"
#define VBT_REC_LOADS 1
#endif
#if VBT_NO_GATHER == 1
#define VBT_GLOAD(W, IDX) "v_mov_b32 " W ", 0\n\t"
#define VBT_GATHERS_N 0
#define VBT_GATHERS_W 0
#elif VBT_NO_GATHER == 2  /* every lane gathers cell 0: the loads are issued and waited for, but they all hit one line */
#define VBT_GLOAD(W, IDX) "buffer_load_sshort " W ", off, %[rs], 0\n\t"
#define VBT_GATHERS_N 1
#define VBT_GATHERS_W 4
#else
#define VBT_GLOAD(W, IDX) "buffer_load_sshort " W ", " IDX ", %[rs], 0 idxen\n\t"
#define VBT_GATHERS_N 1
#define VBT_GATHERS_W 4
#endif
// loads that may stay in flight at the top of an iteration: those of the iteration before it (narrow or wide: the variant) and of the
// one before that (counted as narrow), each + its record request where the records come from global memory
#if VBT_GATHERS_N == 0
#if VBT_REC_LOADS
#define VBT_VMC_N "2"
#define VBT_VMC_W "2"
#else
#define VBT_VMC_N "0"
#define VBT_VMC_W "0"
#endif
#elif VBT_REC_LOADS
#define VBT_VMC_N "4"
#define VBT_VMC_W "7"
#else
#define VBT_VMC_N "2"
#define VBT_VMC_W "5"
#endif
#define VBT_B1 " src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define VBT_B2 " src0_sel:DWORD src1_sel:BYTE_2\n\t"
#define VBT_B3 " src0_sel:DWORD src1_sel:BYTE_3\n\t"

// developer aids (tools/loop_profile.sh; s_memtime stamps the moment it issues):
//   -DVBT_LOOP_PROF=1: cycles an iteration is parked at its two waits, summed in s[64:65] (the loads: gathers + record) and s[66:67]
//                      (the LDS reads);
//   -DVBT_LOOP_PROF=2: whole iterations by kind -- s[78:79] / s80 cycles and count of the iterations that took the common pass,
//                      s[82:83] / s81 of the others (an iteration's two stamps are summed behind the NEXT iteration's LDS wait).
// The numbers are left in LDS at %[plds] behind the loop.
#if VBT_LOOP_PROF == 1
#define VBT_PROF_WAIT(ACC_LO, ACC_HI, WAIT)                                                          \
    "s_memtime s[62:63]\n\t" WAIT "s_memtime s[68:69]\n\ts_waitcnt lgkmcnt(0)\n\t"                   \
    "s_sub_u32 s62, s68, s62\n\ts_subb_u32 s63, s69, s63\n\t"                                        \
    "s_add_u32 " ACC_LO ", " ACC_LO ", s62\n\ts_addc_u32 " ACC_HI ", " ACC_HI ", s63\n\t"
#define VBT_PROF_INIT "s_mov_b64 s[64:65], 0\n\ts_mov_b64 s[66:67], 0\n\t"
#define VBT_PROF_OUT "v_mov_b32 v60, s64\n\tv_mov_b32 v61, s65\n\tv_mov_b32 v62, s66\n\tv_mov_b32 v63, s67\n\t"          \
                     "ds_write_b64 %[plds], v[60:61]\n\tds_write_b64 %[plds], v[62:63] offset:8\n\ts_waitcnt lgkmcnt(0)\n\t"
#define VBT_PROF_CLOBBERS "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69",
#define VBT_PROF2_START
#define VBT_PROF2_ACC(KIND, THIS)
#define VBT_PROF2_END
#elif VBT_LOOP_PROF == 2
#define VBT_PROF_WAIT(ACC_LO, ACC_HI, WAIT) WAIT
#define VBT_PROF_INIT "s_mov_b64 s[70:71], 0\n\ts_mov_b64 s[72:73], 0\n\ts_mov_b32 s74, 0\n\ts_mov_b64 s[78:79], 0\n\ts_mov_b64 s[82:83], 0\n\t"  \
                      "s_mov_b32 s80, 0\n\ts_mov_b32 s81, 0\n\t"
#define VBT_PROF2_START "s_memtime s[76:77]\n\t"
/* behind the iteration's LDS wait: the previous iteration's stamps (s[70:71] start, s[72:73] end, s74 its kind) are in; then this one's */
#define VBT_PROF2_ACC(KIND, THIS)                                                                    \
    "s_sub_u32 s62, s72, s70\n\ts_subb_u32 s63, s73, s71\n\t"                                        \
    "s_cmp_eq_u32 s74, 0\n\ts_cbranch_scc0 .LBBvbt_pg%=_" KIND "\n\t"                                \
    "s_add_u32 s78, s78, s62\n\ts_addc_u32 s79, s79, s63\n\ts_add_u32 s80, s80, 1\n\ts_branch .LBBvbt_pe%=_" KIND "\n"  \
    "\n.LBBvbt_pg%=_" KIND ":\n\t"                                                                    \
    "s_add_u32 s82, s82, s62\n\ts_addc_u32 s83, s83, s63\n\ts_add_u32 s81, s81, 1\n"                 \
    "\n.LBBvbt_pe%=_" KIND ":\n\t"                                                                    \
    "s_mov_b64 s[70:71], s[76:77]\n\ts_mov_b32 s74, " THIS "\n\t"
#define VBT_PROF2_END "s_memtime s[72:73]\n\t"
#define VBT_PROF_OUT "v_mov_b32 v60, s78\n\tv_mov_b32 v61, s79\n\tv_mov_b32 v62, s82\n\tv_mov_b32 v63, s83\n\tv_mov_b32 v64, s80\n\tv_mov_b32 v65, s81\n\t"  \
                     "ds_write_b64 %[plds], v[60:61]\n\tds_write_b64 %[plds], v[62:63] offset:8\n\tds_write_b64 %[plds], v[64:65] offset:16\n\ts_waitcnt lgkmcnt(0)\n\t"
#define VBT_PROF_CLOBBERS "s62", "s63", "s70", "s71", "s72", "s73", "s74", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
#else
#define VBT_PROF_WAIT(ACC_LO, ACC_HI, WAIT) WAIT
#define VBT_PROF_INIT
#define VBT_PROF_OUT
#define VBT_PROF_CLOBBERS
#define VBT_PROF2_START
#define VBT_PROF2_ACC(KIND, THIS)
#define VBT_PROF2_END
#endif

// combine the four phases of every candidate and write its node: in KLO / KHI the lane's (field | right id, cost + connection cost),
// VM = lanes that saw a predecessor (and write).  Minimum cost over the quad, then the smallest field among the lanes that hold it
// (the last inserted predecessor, lattice.rs:141-146); + word cost (lattice.rs:125); cost -> the slot record, field -> the low half
// of the candidate record (the back pointer).  FILL1/FILL2: two independent instructions for the DPP wait states.
// (VBT_VT: the masked cost while the quad minimum is taken -- dead before the field goes into v70, so with the records in LDS it IS v70;
// VBT_VNC: the node's cost.  Register layout of the build with the records in LDS: v24-v79, see VBT_SWEEP_CLOBBERS.)
#if VBT_LDS_REC
#define VBT_VT "v70"
#define VBT_VNC "v24"
#else
#define VBT_VT "v73"
#define VBT_VNC "v72"
#endif
#define VBT_FINISH(KLO, KHI, VM, CA, FILL1, FILL2)                                                   \
    "v_cndmask_b32_e64 " VBT_VT ", -1, " KHI ", " VM "\n\t"                                          \
    FILL1 FILL2                                                                                       \
    "v_min_u32_dpp v69, " VBT_VT ", " VBT_VT VBT_DPP1                                                 \
    "v_add_u32_sdwa v71, v68, %[offk]" VBT_SDWA_LO                                                    \
    "s_nop 0\n\t"                                                                                     \
    "v_min_u32_dpp v69, v69, v69" VBT_DPP2                                                            \
    "v_cmp_eq_u32_e32 vcc, " VBT_VT ", v69\n\t"                                                       \
    "v_add_u32_sdwa " VBT_VNC ", v69, sext(v68)" VBT_SDWA_SEXT_HI                                     \
    "v_cndmask_b32_e32 v70, -1, " KLO ", vcc\n\t"                                                     \
    "s_nop 1\n\t"                                                                                     \
    "v_min_u32_dpp v70, v70, v70" VBT_DPP1                                                            \
    "s_nop 1\n\t"                                                                                     \
    "v_min_u32_dpp v70, v70, v70" VBT_DPP2                                                            \
    "s_mov_b64 exec, " VM "\n\t"                                                                      \
    "ds_write_b32 v71, " VBT_VNC " offset:4\n\t"                                                      \
    "ds_write_b16_d16_hi " CA ", v70\n\t"

// one unit of a general pass: (cost + connection cost, field) of predecessor 4 i + k against the running minimum; MASK = "" (a unit
// in front of the last: full for every candidate that exists -- lanes of the others compute garbage nobody writes) or the lanes
// that hold a pair in this, the last, unit: KREG = k + 4 i against the predecessors of the round, s[50:51] = cl < candidates
#define VBT_UNIT(KLO, KHI, W, MASK)                                                                  \
    "v_add_u32 v" KHI ", v" KHI ", " W "\n\t"                                                         \
    "v_cmp_lt_u64_e32 vcc, v[" KLO ":" KHI "], v[74:75]\n\t"                                          \
    MASK                                                                                              \
    "v_cndmask_b32_e32 v74, v74, v" KLO ", vcc\n\t"                                                   \
    "v_cndmask_b32_e32 v75, v75, v" KHI ", vcc\n\t"
#define VBT_LAST(KREG, META)                                                                         \
    "v_cmp_lt_u32_sdwa s[52:53], " KREG ", " META VBT_B1                                              \
    "s_and_b64 s[52:53], s[52:53], s[50:51]\n\t"
#define VBT_LASTMASK "s_and_b64 vcc, vcc, s[52:53]\n\ts_nop 0\n\t"

// the issue side of a pass up to the mask of its unit 0: addresses, left row, right id of predecessor k, flags -> FL, lanes of unit
// 0 -> s[48:49]; SCC = a wide pass (s46 != 0).  R0 / R1 = the record's two words.
#define VBT_ISSUE_HEAD(PA, NCA, R0, R1, FL)                                                          \
    "v_add_u32_sdwa " PA ", " R0 ", v30" VBT_SDWA_LO                                                  \
    "v_add_u32_sdwa " NCA ", " R1 ", v31" VBT_SDWA_LO                                                 \
    "ds_read_b32 v54, " NCA "\n\t"                                                                    \
    "ds_read_b32 v55, " PA "\n\t"                                                                     \
    "v_readfirstlane_b32 s45, " R1 "\n\t"                                                             \
    "v_cmp_lt_u32_sdwa vcc, v25, " R0 VBT_B3                                                          \
    "v_cmp_lt_u32_sdwa s[52:53], v26, " R1 VBT_B3                                                     \
    "s_bfe_u32 " FL ", s45, 0x80010\n\t"                                                              \
    "s_and_b64 s[48:49], vcc, s[52:53]\n\t"                                                           \
    "s_and_b32 s46, " FL ", 6\n\t"
#define VBT_WIDE_READS(PA)                                                                           \
    "ds_read_b32 v56, " PA " offset:32\n\t"                                                           \
    "ds_read_b32 v57, " PA " offset:64\n\t"                                                           \
    "ds_read_b32 v58, " PA " offset:96\n\t"
// units 1..3 of a wide pass behind unit 0's gather: cell indices, lanes, gathers
#define VBT_WIDE(W1, W2, W3, R0, R1)                                                                 \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_add_u32_sdwa v56, v56, v54" VBT_SDWA_LO                                                        \
    "v_add_u32_sdwa v57, v57, v54" VBT_SDWA_LO                                                        \
    "v_add_u32_sdwa v58, v58, v54" VBT_SDWA_LO                                                        \
    "v_cmp_lt_u32_sdwa s[50:51], v25, " R0 VBT_B3                                                     \
    "v_cmp_lt_u32_sdwa s[52:53], v27, " R1 VBT_B3                                                     \
    "v_cmp_lt_u32_sdwa s[54:55], v28, " R1 VBT_B3                                                     \
    "v_cmp_lt_u32_sdwa s[56:57], v29, " R1 VBT_B3                                                     \
    "s_and_b64 exec, s[50:51], s[52:53]\n\t"                                                          \
    VBT_GLOAD(W1, "v56")                                                                            \
    "s_and_b64 exec, s[50:51], s[54:55]\n\t"                                                          \
    VBT_GLOAD(W2, "v57")                                                                            \
    "s_and_b64 exec, s[50:51], s[56:57]\n\t"                                                          \
    VBT_GLOAD(W3, "v58")                                                                            \
    "s_mov_b64 exec, -1\n\t"
// meta of the pass being issued, for the slot: predecessors of the round << 8 | phases << 16 | candidates << 24 (bytes 2, 3 of
// the first record word over bytes 2, 3 of the second; s58 = the byte selector)
#define VBT_META(META, R0, R1) "v_perm_b32 " META ", " R0 ", " R1 ", s58\n\t"

// One iteration, in line: the common pass.  U = iteration mod 6, V = variant (N / W), VMC = loads that may stay in flight,
// W0..W3 PA CA META M FL = the gather slot's registers, R0 R1 = the record of the pass to issue, RLOAD / OFF = slot and byte offset
// of the record to request, TAILN = what follows a narrow issue.
#define VBT_ITER(U, V, VMC, W0, W1, W2, W3, PA, CA, META, M, FL, R0, R1, RLOAD, OFF, TAILN, TAILW)                          \
    "\n.LBBvbt_i" U V "_%=:\n\t"                                                                      \
    VBT_PROF_WAIT("s64", "s65", "s_waitcnt vmcnt(" VMC ")\n\t")  /* the gathers of the pass in hand, the record of the pass to issue */ \
    VBT_PROF2_START                                                                                   \
    "ds_read_b64 v[60:61], " PA "\n\t"                           /* predecessor k of the pass in hand */ \
    "ds_read_b32 v68, " CA " offset:4\n\t"                       /* its candidate: slot offset | word cost */ \
    "s_cmp_lg_u32 " FL ", 25\n\t"                                /* one unit that starts and ends the step? */ \
    "s_cbranch_scc1 .LBBvbt_g" U V "_%=\n\t"                                                          \
    VBT_ISSUE_HEAD(PA, "v53", R0, R1, FL)                                                             \
    "s_cbranch_scc1 .LBBvbt_r" U V "_%=\n"                                                            \
    "\n.LBBvbt_b" U V "_%=:\n\t"                                                                      \
    VBT_PROF_WAIT("s66", "s67", "s_waitcnt lgkmcnt(0)\n\t")                                           \
    VBT_PROF2_ACC("c" U V, "0")                                                                             \
    VBT_RECLOAD(RLOAD, OFF)                                      /* the record of six passes on */ \
    "v_add_u32 v61, v61, " W0 "\n\t"                             /* wrapping i32 add of the connection cost (lattice.rs:139) */ \
    VBT_FINISH("v60", "v61", M, CA, "v_add_u32_sdwa v55, v55, v54" VBT_SDWA_LO, VBT_META(META, R0, R1))                      \
    "\n.LBBvbt_j" U V "_%=:\n\t"                                                                      \
    "s_mov_b64 exec, s[48:49]\n\t"                                                                    \
    VBT_GLOAD(W0, "v55")                                                                            \
    "s_cmp_lg_u32 s46, 0\n\t"                                    /* more units: a wide pass */        \
    "s_cbranch_scc1 .LBBvbt_w" U V "_%=\n\t"                                                          \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_mov_b32 " CA ", v53\n\t"                                                                       \
    "s_mov_b64 " M ", s[48:49]\n\t"                                                                   \
    VBT_PROF2_END                                                                                     \
    TAILN

// the out-of-line blocks of an iteration: the general consume side (any number of units, rounds of a step, empty passes), the right
// ids of a wide pass, its gathers
#define VBT_ITER_OOL(U, V, VMC, W0, W1, W2, W3, PA, CA, META, M, FL, R0, R1, RLOAD, OFF, TAILN, TAILW)                      \
    "\n.LBBvbt_g" U V "_%=:\n\t"                                                                      \
    "ds_read_b64 v[62:63], " PA " offset:32\n\t"                                                      \
    "ds_read_b64 v[64:65], " PA " offset:64\n\t"                                                      \
    "ds_read_b64 v[66:67], " PA " offset:96\n\t"                                                      \
    "s_mov_b32 s47, " FL "\n\t"                                  /* flags of the pass in hand */      \
    VBT_ISSUE_HEAD(PA, "v53", R0, R1, FL)                                                             \
    "s_cbranch_scc0 .LBBvbt_n" U V "_%=\n\t"                                                          \
    VBT_WIDE_READS(PA)                                                                                \
    "\n.LBBvbt_n" U V "_%=:\n\t"                                                                      \
    VBT_PROF_WAIT("s66", "s67", "s_waitcnt lgkmcnt(0)\n\t")                                           \
    VBT_PROF2_ACC("g" U V, "1")                                                                             \
    VBT_RECLOAD(RLOAD, OFF)                                                                           \
    "v_add_u32_sdwa v55, v55, v54" VBT_SDWA_LO                                                        \
    "v_cmp_lt_u32_sdwa s[50:51], v25, " META VBT_B3              /* candidates of the pass in hand */ \
    "s_and_b32 s45, s47, 7\n\t"                                  /* units (0: an empty pass -- unit 0 under no lanes) */ \
    "s_cmp_lt_u32 s45, 2\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l0" U V "_%=\n\t"                                                         \
    VBT_UNIT("60", "61", W0, "")                                                                      \
    "s_cmp_lt_u32 s45, 3\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l1" U V "_%=\n\t"                                                         \
    VBT_UNIT("62", "63", W1, "")                                                                      \
    "s_cmp_lt_u32 s45, 4\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_l2" U V "_%=\n\t"                                                         \
    VBT_UNIT("64", "65", W2, "")                                                                      \
    VBT_LAST("v29", META)                                                                             \
    VBT_UNIT("66", "67", W3, VBT_LASTMASK)                                                            \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l0" U V "_%=:\n\t"                                                                     \
    VBT_LAST("v26", META)                                                                             \
    VBT_UNIT("60", "61", W0, VBT_LASTMASK)                                                            \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l1" U V "_%=:\n\t"                                                                     \
    VBT_LAST("v27", META)                                                                             \
    VBT_UNIT("62", "63", W1, VBT_LASTMASK)                                                            \
    "s_branch .LBBvbt_f" U V "_%=\n"                                                                  \
    "\n.LBBvbt_l2" U V "_%=:\n\t"                                                                     \
    VBT_LAST("v28", META)                                                                             \
    VBT_UNIT("64", "65", W2, VBT_LASTMASK)                                                            \
    "\n.LBBvbt_f" U V "_%=:\n\t"                                                                      \
    "s_bitcmp1_b32 s47, 4\n\t"                                   /* the last round of the step: combine, write, start over */ \
    "s_cbranch_scc0 .LBBvbt_m" U V "_%=\n\t"                                                          \
    "v_cmp_lt_u32_sdwa s[52:53], v26, " META VBT_B2              /* phases that saw a predecessor in the step */ \
    "s_and_b64 s[54:55], s[50:51], s[52:53]\n\t"                                                      \
    VBT_FINISH("v74", "v75", "s[54:55]", CA, "s_nop 0\n\t", "s_nop 0\n\t")                            \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "v_mov_b32 v74, -1\n\t"                                                                           \
    "v_mov_b32 v75, -1\n\t"                                                                           \
    "\n.LBBvbt_m" U V "_%=:\n\t"                                                                      \
    VBT_META(META, R0, R1)                                                                            \
    "s_branch .LBBvbt_j" U V "_%=\n"                                                                  \
    "\n.LBBvbt_r" U V "_%=:\n\t"                                                                      \
    VBT_WIDE_READS(PA)                                                                                \
    "s_branch .LBBvbt_b" U V "_%=\n"                                                                  \
    "\n.LBBvbt_w" U V "_%=:\n\t"                                                                      \
    VBT_WIDE(W1, W2, W3, R0, R1)                                                                      \
    "v_mov_b32 " CA ", v53\n\t"                                                                       \
    "s_mov_b64 " M ", s[48:49]\n\t"                                                                   \
    VBT_PROF2_END                                                                                     \
    TAILW

// gather slots and record slots
#define VBT_G0 "v32", "v33", "v34", "v35", "v44", "v45", "v46", "s[36:37]", "s42"
#define VBT_G1 "v36", "v37", "v38", "v39", "v47", "v48", "v49", "s[38:39]", "s43"
#define VBT_G2 "v40", "v41", "v42", "v43", "v50", "v51", "v52", "s[40:41]", "s44"
#if VBT_LDS_REC
// records in LDS: three slots, record r in slot r mod 3 -- the same index as its pass's gather slot.  Iteration si issues from record
// si + 3 and requests record si + 5 into the slot of record si + 2 (dead since the issue of pass si + 2, an iteration ago): a request
// sits behind its iteration's LDS wait and has landed at the next iteration's, so it is first touched two iterations on.  (Round 5
// requested six ahead into six slots -- what the records needed when they were vector loads from global memory; three slots and
// v73 / v72 folded away bring the block from v24-v87 to v24-v79: 80 VGPRs, six waves per SIMD for the lean instance.)
#define VBT_R0 "v76", "v77"
#define VBT_R1 "v78", "v79"
#define VBT_R2 "v72", "v73"
#define VBT_R0P "v[76:77]"
#define VBT_R1P "v[78:79]"
#define VBT_R2P "v[72:73]"
#else
#define VBT_R0 "v76", "v77"
#define VBT_R1 "v78", "v79"
#define VBT_R2 "v80", "v81"
#define VBT_R3 "v82", "v83"
#define VBT_R4 "v84", "v85"
#define VBT_R5 "v86", "v87"
#endif
// behind iterations 1 and 3: two passes done -- out if none are left; behind iteration 5 the records move on by six
#define VBT_HALF(NEXTLABEL)                                                                          \
    "s_sub_i32 s59, s59, 2\n\t"                                                                       \
    "s_cmp_gt_i32 s59, 0\n\t"                                                                         \
    "s_cbranch_scc1 " NEXTLABEL "_%=\n\t"                                                             \
    "s_branch .LBBvbt_x_%=\n"
#define VBT_HALF_FALL                                                                                \
    "s_sub_i32 s59, s59, 2\n\t"                                                                       \
    "s_cmp_gt_i32 s59, 0\n\t"                                                                         \
    "s_cbranch_scc0 .LBBvbt_x_%=\n\t"
#if VBT_LDS_REC
#define VBT_TRIP_ADVANCE "v_add_u32 v59, 48, v59\n\t"
#else
#define VBT_TRIP_ADVANCE "s_add_u32 s60, s60, 48\n\ts_addc_u32 s61, s61, 0\n\t"
#endif
#define VBT_TRIP(NEXT)                                                                               \
    VBT_TRIP_ADVANCE                                                                                  \
    "s_sub_i32 s59, s59, 2\n\t"                                                                       \
    "s_cmp_gt_i32 s59, 0\n\t"                                                                         \
    "s_cbranch_scc1 .LBBvbt_i0" NEXT "_%=\n\t"                                                        \
    "s_branch .LBBvbt_x_%=\n"

#define VBT_EXPAND(M, ...) M(__VA_ARGS__)
#if VBT_LDS_REC
// iteration U: gather slot and record slot U mod 3; requests record U + 5 of the trip (byte offset 8 (U + 5)) into slot (U + 2) mod 3
#define VBT_IT0(M, V, VMC, TN, TW) VBT_EXPAND(M, "0", V, VMC, VBT_G0, VBT_R0, VBT_R2P, "40", TN, TW)
#define VBT_IT1(M, V, VMC, TN, TW) VBT_EXPAND(M, "1", V, VMC, VBT_G1, VBT_R1, VBT_R0P, "48", TN, TW)
#define VBT_IT2(M, V, VMC, TN, TW) VBT_EXPAND(M, "2", V, VMC, VBT_G2, VBT_R2, VBT_R1P, "56", TN, TW)
#define VBT_IT3(M, V, VMC, TN, TW) VBT_EXPAND(M, "3", V, VMC, VBT_G0, VBT_R0, VBT_R2P, "64", TN, TW)
#define VBT_IT4(M, V, VMC, TN, TW) VBT_EXPAND(M, "4", V, VMC, VBT_G1, VBT_R1, VBT_R0P, "72", TN, TW)
#define VBT_IT5(M, V, VMC, TN, TW) VBT_EXPAND(M, "5", V, VMC, VBT_G2, VBT_R2, VBT_R1P, "80", TN, TW)

// the prologue's issue of pass P (record slot P, gather slot P); the record of pass P + 3 is requested into the same slot behind it
#define VBT_PRO(P, W0, W1, W2, W3, PA, CA, META, M, FL, R0, R1)                                      \
    VBT_ISSUE_HEAD(PA, CA, R0, R1, FL)                                                                \
    "s_cbranch_scc0 .LBBvbt_pn" P "_%=\n\t"                                                           \
    VBT_WIDE_READS(PA)                                                                                \
    "\n.LBBvbt_pn" P "_%=:\n\t"                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    "v_add_u32_sdwa v55, v55, v54" VBT_SDWA_LO                                                        \
    VBT_META(META, R0, R1)                                                                            \
    "s_mov_b64 " M ", s[48:49]\n\t"                                                                   \
    "s_mov_b64 exec, s[48:49]\n\t"                                                                    \
    VBT_GLOAD(W0, "v55")                                                                            \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "s_cmp_lg_u32 s46, 0\n\t"
#define VBT_PRO0 VBT_EXPAND(VBT_PRO, "0", VBT_G0, VBT_R0)
#define VBT_PRO1 VBT_EXPAND(VBT_PRO, "1", VBT_G1, VBT_R1)
#define VBT_PRO2 VBT_EXPAND(VBT_PRO, "2", VBT_G2, VBT_R2)
#define VBT_PRO_REC0 VBT_RECLOAD(VBT_R0P, "24")
#define VBT_PRO_REC1 VBT_RECLOAD(VBT_R1P, "32")
#define VBT_HEAD_R0 VBT_R0P
#define VBT_HEAD_R1 VBT_R1P
#define VBT_HEAD_R2 VBT_R2P
#else
// iteration U: gather slot U mod 3, issues from record slot (U + 3) mod 6, requests into record slot U
#define VBT_IT0(M, V, VMC, TN, TW) VBT_EXPAND(M, "0", V, VMC, VBT_G0, VBT_R3, "v[76:77]", "48", TN, TW)
#define VBT_IT1(M, V, VMC, TN, TW) VBT_EXPAND(M, "1", V, VMC, VBT_G1, VBT_R4, "v[78:79]", "56", TN, TW)
#define VBT_IT2(M, V, VMC, TN, TW) VBT_EXPAND(M, "2", V, VMC, VBT_G2, VBT_R5, "v[80:81]", "64", TN, TW)
#define VBT_IT3(M, V, VMC, TN, TW) VBT_EXPAND(M, "3", V, VMC, VBT_G0, VBT_R0, "v[82:83]", "72", TN, TW)
#define VBT_IT4(M, V, VMC, TN, TW) VBT_EXPAND(M, "4", V, VMC, VBT_G1, VBT_R1, "v[84:85]", "80", TN, TW)
#define VBT_IT5(M, V, VMC, TN, TW) VBT_EXPAND(M, "5", V, VMC, VBT_G2, VBT_R2, "v[86:87]", "88", TN, TW)

// the prologue's issue of pass P (record slot P, gather slot P): the record of pass P + 3 is requested first
#define VBT_PRO_(P, W0, W1, W2, W3, PA, CA, META, M, FL, R0, R1, RLOAD, OFF)                         \
    VBT_RECLOAD(RLOAD, OFF)                                                                           \
    VBT_ISSUE_HEAD(PA, CA, R0, R1, FL)                                                                \
    "s_cbranch_scc0 .LBBvbt_pn" P "_%=\n\t"                                                           \
    VBT_WIDE_READS(PA)                                                                                \
    "\n.LBBvbt_pn" P "_%=:\n\t"                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    "v_add_u32_sdwa v55, v55, v54" VBT_SDWA_LO                                                        \
    VBT_META(META, R0, R1)                                                                            \
    "s_mov_b64 " M ", s[48:49]\n\t"                                                                   \
    "s_mov_b64 exec, s[48:49]\n\t"                                                                    \
    VBT_GLOAD(W0, "v55")                                                                            \
    "s_mov_b64 exec, -1\n\t"                                                                          \
    "s_cmp_lg_u32 s46, 0\n\t"
#define VBT_PRO0 VBT_EXPAND(VBT_PRO_, "0", VBT_G0, VBT_R0, "v[82:83]", "24")
#define VBT_PRO1 VBT_EXPAND(VBT_PRO_, "1", VBT_G1, VBT_R1, "v[84:85]", "32")
#define VBT_PRO2 VBT_EXPAND(VBT_PRO_, "2", VBT_G2, VBT_R2, "v[86:87]", "40")
#define VBT_PRO_REC0
#define VBT_PRO_REC1
#define VBT_HEAD_R0 "v[76:77]"
#define VBT_HEAD_R1 "v[78:79]"
#define VBT_HEAD_R2 "v[80:81]"
#endif

#if VBT_LDS_REC
#define VBT_REC_BASE "v_mov_b32 v59, %[rp]\n\t"
#define VBT_REC_HEAD "v59"
#define VBT_ZERO_INIT
#else
#define VBT_REC_BASE "s_mov_b64 s[60:61], %[rp]\n\t"
#define VBT_REC_HEAD "%[hd]"
#define VBT_ZERO_INIT "v_mov_b32 v24, 0\n\t"
#endif
#define VBT_SWEEP_TEXT                                                                               \
    /* Nothing the compiler issued may still be in flight: a load whose result no lane went on to use (the candidate records */ \
    /* requested ahead of the reachability sweep, for lanes behind the last candidate) is never waited for by compiled code, */ \
    /* and its destination may be one of the registers this block owns -- it would land in the middle of the loop.  (Round 4  */ \
    /* had this wait by accident: the drain of the record stores.  Found in round 5 as wrong tokens in one build variant.)   */ \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                               \
    VBT_ZERO_INIT                                                                                     \
    "v_lshrrev_b32 v25, 2, %[ln]\n\t"                                                                 \
    "v_and_b32 v26, 3, %[ln]\n\t"                                                                     \
    "v_add_u32 v27, 4, v26\n\t"                                                                       \
    "v_add_u32 v28, 8, v26\n\t"                                                                       \
    "v_add_u32 v29, 12, v26\n\t"                                                                      \
    "v_lshlrev_b32 v30, 3, v26\n\t"                                                                   \
    "v_lshlrev_b32 v31, 3, v25\n\t"                                                                   \
    "v_mov_b32 v74, -1\n\t"                                                                           \
    "v_mov_b32 v75, -1\n\t"                                                                           \
    VBT_REC_BASE                                                                                      \
    "s_mov_b32 s59, %[sl]\n\t"                                                                        \
    "s_mov_b32 s58, 0x07060302\n\t"                                                                   \
    VBT_PROF_INIT                                                                                     \
    /* the first three records out of LDS (global records: the builder left a copy there, no round trip at the start) */ \
    "ds_read_b64 " VBT_HEAD_R0 ", " VBT_REC_HEAD "\n\t"                                               \
    "ds_read_b64 " VBT_HEAD_R1 ", " VBT_REC_HEAD " offset:8\n\t"                                      \
    "ds_read_b64 " VBT_HEAD_R2 ", " VBT_REC_HEAD " offset:16\n\t"                                     \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                        \
    /* prologue: the gathers of passes 0, 1 and 2 (and the requests for the records of the passes behind them) */ \
    VBT_PRO0                                                                                          \
    "s_cbranch_scc0 .LBBvbt_p0_%=\n\t"                                                                \
    VBT_EXPAND(VBT_WIDE, "v33", "v34", "v35", VBT_R0)                                                 \
    "\n.LBBvbt_p0_%=:\n\t"                                                                            \
    VBT_PRO_REC0                                                                                      \
    VBT_PRO1                                                                                          \
    "s_cbranch_scc0 .LBBvbt_p1_%=\n\t"                                                                \
    VBT_EXPAND(VBT_WIDE, "v37", "v38", "v39", VBT_R1)                                                 \
    "\n.LBBvbt_p1_%=:\n\t"                                                                            \
    VBT_PRO_REC1                                                                                      \
    VBT_PRO2                                                                                          \
    "s_cbranch_scc0 .LBBvbt_i0N_%=\n\t"                                                               \
    VBT_EXPAND(VBT_WIDE, "v41", "v42", "v43", VBT_R2)                                                 \
    "s_branch .LBBvbt_i0W_%=\n"                                                                       \
    /* the loop: the narrow variants in line */                                                       \
    VBT_IT0(VBT_ITER, "N", VBT_VMC_N, "", "")                                                               \
    VBT_IT1(VBT_ITER, "N", VBT_VMC_N, VBT_HALF_FALL, "")                                                    \
    VBT_IT2(VBT_ITER, "N", VBT_VMC_N, "", "")                                                               \
    VBT_IT3(VBT_ITER, "N", VBT_VMC_N, VBT_HALF_FALL, "")                                                    \
    VBT_IT4(VBT_ITER, "N", VBT_VMC_N, "", "")                                                               \
    VBT_IT5(VBT_ITER, "N", VBT_VMC_N, VBT_TRIP("N"), "")                                                    \
    VBT_IT0(VBT_ITER, "W", VBT_VMC_W, "s_branch .LBBvbt_i1N_%=\n", "")                                      \
    VBT_IT1(VBT_ITER, "W", VBT_VMC_W, VBT_HALF(".LBBvbt_i2N"), "")                                          \
    VBT_IT2(VBT_ITER, "W", VBT_VMC_W, "s_branch .LBBvbt_i3N_%=\n", "")                                      \
    VBT_IT3(VBT_ITER, "W", VBT_VMC_W, VBT_HALF(".LBBvbt_i4N"), "")                                          \
    VBT_IT4(VBT_ITER, "W", VBT_VMC_W, "s_branch .LBBvbt_i5N_%=\n", "")                                      \
    VBT_IT5(VBT_ITER, "W", VBT_VMC_W, VBT_TRIP("N"), "")                                                    \
    VBT_IT0(VBT_ITER_OOL, "N", VBT_VMC_N, "", "s_branch .LBBvbt_i1W_%=\n")                                  \
    VBT_IT1(VBT_ITER_OOL, "N", VBT_VMC_N, "", VBT_HALF(".LBBvbt_i2W"))                                      \
    VBT_IT2(VBT_ITER_OOL, "N", VBT_VMC_N, "", "s_branch .LBBvbt_i3W_%=\n")                                  \
    VBT_IT3(VBT_ITER_OOL, "N", VBT_VMC_N, "", VBT_HALF(".LBBvbt_i4W"))                                      \
    VBT_IT4(VBT_ITER_OOL, "N", VBT_VMC_N, "", "s_branch .LBBvbt_i5W_%=\n")                                  \
    VBT_IT5(VBT_ITER_OOL, "N", VBT_VMC_N, "", VBT_TRIP("W"))                                                \
    VBT_IT0(VBT_ITER_OOL, "W", VBT_VMC_W, "", "s_branch .LBBvbt_i1W_%=\n")                                  \
    VBT_IT1(VBT_ITER_OOL, "W", VBT_VMC_W, "", VBT_HALF(".LBBvbt_i2W"))                                      \
    VBT_IT2(VBT_ITER_OOL, "W", VBT_VMC_W, "", "s_branch .LBBvbt_i3W_%=\n")                                  \
    VBT_IT3(VBT_ITER_OOL, "W", VBT_VMC_W, "", VBT_HALF(".LBBvbt_i4W"))                                      \
    VBT_IT4(VBT_ITER_OOL, "W", VBT_VMC_W, "", "s_branch .LBBvbt_i5W_%=\n")                                  \
    VBT_IT5(VBT_ITER_OOL, "W", VBT_VMC_W, "", VBT_TRIP("W"))                                                \
    "\n.LBBvbt_x_%=:\n\t"                                                                             \
    "s_waitcnt vmcnt(0)\n\t"                                                                          \
    "s_waitcnt lgkmcnt(0)\n\t"           /* (record reads of the empty passes behind the last: their registers go back to the compiler) */ \
    VBT_PROF_OUT                                                                                      \
    "s_mov_b64 exec, -1"

#if VBT_LDS_REC
#define VBT_CLOBBERS_HI
#else
#define VBT_CLOBBERS_HI "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87",
#endif
#define VBT_SWEEP_CLOBBERS                                                                           \
    "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",                \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",   \
    "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",          \
    "v76", "v77", "v78", "v79", VBT_CLOBBERS_HI                                                                                    \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",               \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", VBT_PROF_CLOBBERS "vcc", "scc", "memory"
