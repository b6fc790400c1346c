// les_march_lab.h -- measurement switches of les_march_kernel.  Included ONLY by -DLES_MARCH_LAB builds (tools/build_variant.sh,
// tools/role_time.sh, tools/phase_probe.py); the product build of les_march.h defines every hook below as a no-op, so none of
// this is in the shipped library.  The outputs of an ablated build are meaningless: the time shows what a resource costs.
//
//   -DLES_MARCH_ROLE_MASK=m    bit 0 / 1 / 2 = compile role A / C / D in; the waves of the other roles exit at once
//   -DLES_MARCH_ROLE_ORDER=o   which role gets the oldest waves of a job slot (index into kRoleOf)
//   -DLES_MARCH_EXP=bits       1 role C issues no statistics loads, 2 role C reads no LDS, 4 role C loads the statistics of image row 0 for every row, 16 role D reads no LDS,
//                              64 role A loads nothing, 128 role D loads / stores nothing
//   -DLES_STATS_POLICY='" nt"' cache-policy bits of the statistics loads (" nt", " sc0", " sc1", " sc0 sc1")
//   -DLES_VOL_NT               streaming (non-temporal) loads of volume / guide rows
//   -DLES_PHASE_TIMING         lane 0 of every wave accumulates the cycles it computes per tick and the cycles it waits at the tick
//                              barrier: les_dbg[2 role] += compute, les_dbg[2 role + 1] += wait, les_dbg[6 + role] += ticks,
//                              les_dbg[9 + role] += the part of `compute` before LES_TICK_MARK (the row loop)   (tools/phase_probe.py)
#pragma once

#ifndef LES_MARCH_ROLE_MASK
#define LES_MARCH_ROLE_MASK 7
#endif
#ifndef LES_MARCH_ROLE_ORDER
#define LES_MARCH_ROLE_ORDER 5
#endif
#ifndef LES_MARCH_EXP
#define LES_MARCH_EXP 0
#endif
#ifndef LES_STATS_POLICY
#define LES_STATS_POLICY ""
#endif

#define LES_LAB_ROLE_ON(bit) ((LES_MARCH_ROLE_MASK & (bit)) != 0)
#define LES_LAB_ABLATE(bit) ((LES_MARCH_EXP & (bit)) != 0)

#if defined(LES_PHASE_TIMING) && !defined(LES_SIM)
#define LES_TICK_BEGIN() unsigned long long tk_c_ = 0, tk_w_ = 0, tk_n_ = 0, tk_r_ = 0, tk_m_ = 0, tk_t_ = clock64()
#define LES_TICK_MARK() (tk_m_ = clock64())
#define LES_TICK_BARRIER() do { const unsigned long long a_ = clock64(); __syncthreads(); const unsigned long long b_ = clock64(); tk_c_ += a_ - tk_t_; tk_r_ += (tk_m_ > tk_t_ ? tk_m_ : a_) - tk_t_; tk_w_ += b_ - a_; tk_t_ = b_; tk_n_++; } while (0)
#define LES_TICK_END(role_) do { if (lane == 0) { atomicAdd(&les_dbg[2 * (role_)], tk_c_); atomicAdd(&les_dbg[2 * (role_) + 1], tk_w_); atomicAdd(&les_dbg[6 + (role_)], tk_n_); atomicAdd(&les_dbg[9 + (role_)], tk_r_); } } while (0)
#endif
