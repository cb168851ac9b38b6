// hb_experiments.h - switches of the EXPERIMENTS build of the library (-DHB_EXPERIMENTS: `make exp` -> stract_amd/lib/libhyperball_exp.so,
// and the SIMT-interpreter test build).  NOT part of the product: the shipped libhyperball.so compiles none of this in and hb_create
// refuses an hb_options that sets any of these bits (include/hyperball.h documents only what an integrator may set).
//
// What lives here: A/B switches of measured-and-rejected kernel forms (kept so that their measurements stay reproducible and their
// parity tests keep running: DESIGN.md "tried and rejected"), and test hooks that force rarely taken paths at small sizes.
// stract_amd/_lib.py loads the experiments build by itself when a test asks for one of them.
//
//   hb_options.tune[1], bits above the low byte (the low byte - gather unroll - is a product knob):
//     bit  8  0x000100  dense fused node rows with the per-tile estimator / Kahan epilogue instead of the once-per-row one (round 3 A/B)
//     bit 11  0x000800  sweep passes always with the three-launch seed collection / expansion, also in the convergence tail
//     bit 12  0x001000  edge partition without the merge / all-reduce / epilogue pipeline over row ranges
//     bit 13  0x002000  bitmap passes gather slot by slot instead of packing each row's surviving sources first (round 2 form)
//     bit 14  0x004000  staged result download off (hb_finish ships the whole image)
//     bit 15  0x008000  a result snapshot after EVERY pass, whatever the graph's size (tests: small graphs)
//     bit 16  0x010000  a final list of 16 entries (tests: the overflow path)
//     bit 17  0x020000  one snapshot only
//     bit 20  0x100000  hb_run's tail pipeline off
//     bit 21  0x200000  the far tail as one workgroup (hb_tail.hip.h) after a small sweep pass; measured no faster (round 5)
//     bit 22  0x400000  ... after any pass (tests)
//     bit 23  0x800000  hb_begin always writes the whole initial state (round 6 A/B: the lean pass 0 off)
//     bit 24  0x1000000 destination partition, changed-only: 64-byte counters on the wire instead of the 6-bit packing (round 6 A/B)
//     bit 25  0x2000000 the transposed work-row graph by atomic scatter (the form before round 6; today only the out-of-memory fallback)
//     bit 26  0x4000000 sweep passes: a touched row's sources 8 per round (index / bit word / gather each a round trip) instead of all at once
//     bit 27  0x8000000 pass 0: the first hub-chunk level through the generic INIT kernel instead of init_level1_kernel (A/B)
//     bit 28  0x10000000 TIMING PROBE, WRONG RESULTS: the dense fused node rows neither read nor write size[] (16 B per row less state traffic)
//   hb_options.tune[7]  hottest counters staged in LDS by the level-1 dense launch (0 = off, <= 2048; measured slower, round 2)
#pragma once
#include <stdint.h>

#ifdef HB_EXPERIMENTS
#define HB_XBITS(tune1) ((uint32_t)(tune1) & ~0xFFu)
#else
#define HB_XBITS(tune1) 0u
#endif
