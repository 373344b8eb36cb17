// Compile-time switches of the split-MFMA GEMM kernels (conv_nn.hip, wgrad_nt.hip, wgrad_nt3r.hip, pwln.hip), one place.  Each is the measured-fastest
// form; the value in brackets is what a tuning build (tools/build_variant.sh NAME -DSWITCH=value) compares it with, and the comment says which
// measurement chose it.  None changes results beyond fp32 summation order.
#pragma once

#ifndef SSV_NN_XONE
#define SSV_NN_XONE 1        // one-path input prefetch where it measured faster in-step (k = 1 tiles -2 %, 128 x 112 k = 3 tiles -2.5 %; the 64-row and
                             // 96-column k = 3 tiles and every 54-column-halo tile were equal or up to 15 % SLOWER with it and keep the two-form prefetch)
#endif

#define SSV_NN_XBUF(KT, WM, NT) (!((KT) == 3 && (NT) == 6))   // input rows by buffer loads (ssv_buf) or through pointers: in-step, per tile -- the k = 1
                                                             // tiles are 5-12 % faster with buffer loads, the 96-column k = 3 tiles 4-6 % with pointers, the rest equal

#ifndef SSV_NN_HALO_SMALL
#define SSV_NN_HALO_SMALL 16  // k = 3 layers whose taps span at most this many columns run the narrow-halo instantiation (tuning builds: -1 = never)
#endif

// waves per SIMD the register allocation must leave room for (the second __launch_bounds__ argument).  Round 5: the 128 x 112 k = 3 tile with the
// 16-column halo at THREE (168 VGPRs, 6 spilled, three workgroups per CU instead of two): 177.6 -> 170.4 us in-step over its ten launches.
// (The same for the 128 x 96 tile: 32 spilled, 63.2 -> 70.3 us; the 64 x 96 tile at four, 128 VGPRs: 47.6 -> 49.5 us.  Not kept.)
#define SSV_NNB_WAVES(KT, WM, NT, EPI, HW) \
  (((KT) == 1 && (WM) == 2 && (NT) == 4 && (EPI) == 0) || ((KT) == 3 && (WM) == 2 && (NT) == 7 && (EPI) == 0 && (HW) == 16) ? 3 : 2)

#ifndef SSV_NNBW_XROW
#define SSV_NNBW_XROW 1      // (tuning builds: 0 = M = 128 j + 1 on the 16-wave kernel with a fifth row tile, as before)
#endif

#ifndef SSV_NNBW_PARK
#define SSV_NNBW_PARK 1      // (tuning builds: 0 = the output stored straight from the accumulator layout, 64-byte pieces of 16 rows per instruction)
#endif

#ifndef SSV_NT_RING
#define SSV_NT_RING 1         // 0 (tuning builds): the k = 3 weight gradient on gemm_nt_bf3_kernel, as before round 4's ring kernel
#endif

// XR (k = 1): M = 128 j + 1 rows (the 513-channel layers): the tiles cover rows 0 .. M - 2 and row M - 1 of the product is added by the workgroups of
// row tile 0 as fp32 dot products of dH(M - 1, t) with the raw input values every staging thread holds before it splits them -- a fifth row tile
// of MFMAs for ONE row otherwise (30 tiles of 128 x 96 for 24).  It pays only together with RANGE slabs (p.bstep == 0: slab z reduces over the
// z-th of Z equal ranges of the launch's B x tchunks chunks, not over whole batch items): with whole items 24 x 16 workgroups do the same two
// items each as 30 x 16 did, and the launch lasts as long as its slowest workgroup.
#ifndef SSV_NT_XROW
#define SSV_NT_XROW 1        // (tuning builds: 0 = a row tile of its own for the last row and whole-item slabs, as before)
#endif

#ifndef SSV_PWLN_XROW
#define SSV_PWLN_XROW 1      // (tuning builds: 0 = M = 513 on five row blocks per wave, as before)
#endif

#ifndef SSV_PWLN_ROLL
#define SSV_PWLN_ROLL 1      // (tuning builds: 0 = the one weight-fragment set of 4 row blocks per wave re-loaded at the end of the chunk)
#endif

#ifndef SSV_PWLN_PARK
#define SSV_PWLN_PARK 1      // (tuning builds: 0 = `pre` and `y` stored straight from the accumulator layout, as in round 4)
#endif

#ifndef SSV_PWLN_BWD_FUSED
#define SSV_PWLN_BWD_FUSED 1
#endif
#ifndef SSV_PWLN_BK64
#define SSV_PWLN_BK64 1      // (tuning builds: 0 = 32-channel chunks in gemm_pwln_kernel<4,4,..> with the staging by waves 0-3 after the chunk's MFMAs, as in round 5)
#endif
#ifndef SSV_NT3R_ABL
#define SSV_NT3R_ABL 0       // (ablation builds only, results are garbage: bit 0 = the dH fragments of gemm_nt3r_kernel are not split, bit 1 = the input
                             //  rows go into the ring unsplit -- bounds what the in-kernel splits cost under the ring, profiles/round6_nt3r_ablation.txt)
#endif
#ifndef SSV_LSTM_FAST_CELL
#define SSV_LSTM_FAST_CELL 1  // the LSTM cell's sigmoids / tanh on v_exp_f32 + v_rcp_f32 (as the LayerNorm / gate kernels' sigmoid since round 5) instead of expf /
                              // tanhf / IEEE division: config 5 10.51 -> 10.16 ms same box; the embeddings' distance from the float oracle 2.6e-7 -> 1.5e-6
                              // (bar 2e-5; tanh(x) = 1 - 2 / (1 + e^2x) is absolute-accurate to ~1e-7, not relative, for |x| << 1).  0 = the libm forms.
#endif
#ifndef SSV_NT_NOBREAK
#define SSV_NT_NOBREAK 1     // gemm_nt_bf3_kernel multiplies the k-steps past a ragged row end too (the staged input is zero there) instead of branching out of the MFMA
                             // sequence: the branch put an s_waitcnt vmcnt(0) in the middle of every step's MFMAs (hipcc's wait state is merged at the join).
#endif
#ifndef SSV_NT_SPLIT_FIRST
#define SSV_NT_SPLIT_FIRST 1 // gemm_nt_bf3_kernel splits the next chunk's dH before it issues the loads of tile s + 2 (behind them the waits for dH came out as
                             // vmcnt(5) .. vmcnt(0), i.e. for the loads just issued).  Both: the loop's waits are counted again (tools/isa_loop_waits.py, loop summary in
                             // profiles/round6_nt_bf3_waits.txt); in-step 118.0 -> 116.8, 28.1 -> 27.7 us -- two workgroups per CU had been covering most of it.
#endif
#ifndef SSV_LSTM_PRESPLIT
#define SSV_LSTM_PRESPLIT 1  // (tuning builds: 0 = the inference wavefront stages fp32 h and splits it in the consumer, as until round 6)
#endif
#ifndef SSV_LSTM_ABL
#define SSV_LSTM_ABL 0               // ablation builds of the pre-split LSTM wavefront step (wrong results): 1 = no stores in the cell epilogue, 2 = two chunks per tile
#endif
#ifndef SSV_LSTM_X0FOLD
#define SSV_LSTM_X0FOLD 1           // layer 0's input frames pre-split, W_ih x_t as the first K segment of layer 0's product; 0 (tuning builds): the projection of all frames
#endif
#ifndef SSV_LSTM_PRESPLIT_TRAIN
#define SSV_LSTM_PRESPLIT_TRAIN 1   // the training forward (every frame's h, c and gates kept) on the pre-split planes too; 0 (tuning builds): inference only
#endif
