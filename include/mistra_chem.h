/* mistra_chem.h — C ABI of the MI355X-native chemistry integrator (libmistra_chem.so).
 *
 * This is the drop-in boundary for ONE path of the reference model (Mistra-UEA/Mistra): the per-grid-cell KPP
 * Rosenbrock (Ros3) integrator of the gas / aer / tot mechanisms.  Each entry point below names the reference
 * interface it replaces (file:line under the reference's src/).  Plain pointers and sizes only.
 *
 * Semantics shared by the integrate calls (what the reference does for one cell, gas.f:710-773):
 *   input   VAR(NVAR), FIX(NFIX), RCONST(NREACT)   = COMMON /GDATA_x/  C(1:NVAR), C(NVAR+1:NSPEC), RCONST
 *                                                    (gas_Global.h:29-41 | aer_Global.h | tot_Global.h)
 *   options fixed as INTEGRATE_x fixes them: Ros3, RTOL 1e-3, ATOL 1e-25 (scalar), Hstart 1e-3 s, Hmin 0,
 *           Hmax |TOUT-TIN|, FacMin 0.2, FacMax 6, FacRej 0.1, FacSafe 0.9, at most 100000 steps (gas.f:739-746, 950-1043)
 *   output  VAR after integrating from TIN to TOUT; ierr = 1 on success or the negative code of
 *           ros_ErrorMsg_x (gas.f:1474-1509: -6 too many steps, -7 step too small, -8 matrix repeatedly singular);
 *           like the reference, VAR then holds whatever state was reached ("print and continue", gas.f:764-767);
 *           stats = COMMON /Statistics/ for that call: Nfun,Njac,Nstp,Nacc,Nrej,Ndec,Nsol,Nsng (gas.f:913-915).
 * Arrays are cell-major: cell c occupies [c*N, (c+1)*N).  Cells are independent (kpp.f90:4310-4470 loops over k).
 * All functions return 0 on success, non-zero on error (mistra_chem_last_error() gives the text).  There is no CPU
 * fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef MISTRA_CHEM_H
#define MISTRA_CHEM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISTRA_MECH_GAS 0 /* gas.f  : NVAR 102, NFIX 3, NREACT 331, LU_NONZERO 1110  */
#define MISTRA_MECH_AER 1 /* aer.f  : NVAR 257, NFIX 5, NREACT 979, LU_NONZERO 6579  */
#define MISTRA_MECH_TOT 2 /* tot.f  : NVAR 417, NFIX 7, NREACT 1627, LU_NONZERO 13503 */

/* Select the HIP device, load the three mechanism tables (directory: env MISTRA_MECH_DIR, else ../mech next to the
 * library) and upload their kernel schedules.  Replaces nothing in the reference (its tables are compiled in:
 * BLOCK DATA JACOBIAN_SPARSE_DATA_x, gas.f:6718 | aer.f:23480 | tot.f:44435); call once before anything else. */
int mistra_chem_init(int device);

/* The same on several GPUs of the node, for a single-process caller such as the reference's Fortran program (the model is
 * serial, kpp.f90:4168): mistra_chem_integrate then cuts the batch into contiguous blocks of cells, one block and one host
 * thread per device — the layer loop of kpp_driver (kpp.f90:4310-4470) carries nothing from one k to the next, so
 * nothing is exchanged.  device_ids = NULL means devices 0 .. n_devices-1; the first one listed is the primary device
 * (one-cell entry points, mistra_chem_describe).  A device may be listed more than once: every listing is a slot of its
 * own (tables, staging buffers, host thread), so the blocks that share a GPU overlap one block's copies with the other's
 * kernel — and the split can be exercised on a one-GPU box.  Replaces a previous init. */
int mistra_chem_init_devices(int n_devices, const int* device_ids);

/* Number of devices the library is initialised on (0 before init). */
int mistra_chem_device_count(void);

/* Release device memory.  Safe to call more than once. */
void mistra_chem_finalize(void);

/* Sizes of a mechanism (gas_Parameters.h:28-49 and siblings).  Any pointer may be NULL.  Works before init. */
int mistra_chem_dims(int mech, int* nvar, int* nfix, int* nreact, int* lu_nonzero);

/* INTEGRATE_x(TIN,TOUT) for ncell cells, host buffers (gas.f:710 | aer.f:1408 | tot.f:2812; called by x_drive at
 * gas.f:173 | aer.f:217 | tot.f:604).  Copies inputs to the device, integrates, copies results back, synchronous.
 * var_out may alias var_in.  ierr (ncell) and stats (ncell*8) may be NULL. */
int mistra_chem_integrate(int mech, int ncell, const double* var_in, const double* fix, const double* rconst,
                          double tin, double tout, double* var_out, int32_t* ierr, int32_t* stats);

/* The same with what INTEGRATE_x leaves behind per cell besides VAR: t_h (ncell*3, may be NULL) receives, per cell, the exit
 * time (-> TIN, gas.f:769), the last accepted step size (-> STEPMIN, gas.f:770) and the step size H when the integrator
 * returned (the H of ros_ErrorMsg_x's message, gas.f:1506).  This is the call a batched kpp_driver makes: one per
 * mechanism and 10-s step for all layers that run it. */
int mistra_chem_integrate_ex(int mech, int ncell, const double* var_in, const double* fix, const double* rconst,
                             double tin, double tout, double* var_out, int32_t* ierr, int32_t* stats, double* t_h);

/* The same from the rate evaluator's inputs instead of the rate constants (mistra_chem_update_rconst below: env [ncell]
 * [mistra_chem_rates_env_size(mech)], what MISTRA_RATES_ENV_x of shim/mistra_kpp_rates.f90 packs per layer): Update_RCONST_x
 * (gas.f:172) and INTEGRATE_x (gas.f:173) of x_drive in one call, RCONST made on the device — 74 / 330 / 544 doubles per layer go
 * up instead of 331 / 979 / 1627. */
int mistra_chem_integrate_env_ex(int mech, int ncell, const double* var_in, const double* fix, const double* env, double tin,
                                 double tout, double* var_out, int32_t* ierr, int32_t* stats, double* t_h);

/* Same call on device-resident buffers (hipMalloc'ed or torch tensors), asynchronous on `hip_stream` (a hipStream_t of
 * that device; NULL = its default stream).  The call runs on the device the buffers live on, which must be one the
 * library was initialised on.  d_ierr (ncell) and d_stats (ncell*8) are required;
 * d_texit_hexit (ncell*2: what INTEGRATE_x leaves in TIN and STEPMIN, gas.f:769-770) may be NULL. */
int mistra_chem_integrate_device(int mech, int ncell, const double* d_var_in, const double* d_fix,
                                 const double* d_rconst, double tin, double tout, double* d_var_out,
                                 int32_t* d_ierr, int32_t* d_stats, double* d_texit_hexit, void* hip_stream);

/* OPT-IN, NOT the reference's behaviour: the same call with a first step size per cell.  INTEGRATE_x starts
 * every call at Hstart = 1e-3 s (gas.f:743) and works its way up to the step the chemistry allows, ~130 steps per call in
 * cloudy layers; a caller that feeds each cell's last step size (d_texit_hexit[2c+1] of the previous chemistry timestep)
 * back as d_hstart[c] skips that ramp.  Results then differ from the reference's at the level of the integrator's own
 * tolerance (study: profiles/r02_hstart_reuse_study.txt).  d_hstart = NULL, or an entry <= 0, gives the reference's 1e-3. */
int mistra_chem_integrate_device_hstart(int mech, int ncell, const double* d_var_in, const double* d_fix,
                                        const double* d_rconst, double tin, double tout, double* d_var_out,
                                        int32_t* d_ierr, int32_t* d_stats, double* d_texit_hexit,
                                        const double* d_hstart, void* hip_stream);

/* Fortran-callable per-cell entry points with the reference's own signature, `SUBROUTINE INTEGRATE_x(TIN,TOUT)`
 * (REAL*8 by reference; data through COMMON /GDATA_x/).  `gdata` is the address of that COMMON block, laid out
 * C(NSPEC), RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX (gas_Global.h:29-58).  On return VAR
 * is updated in place, *tin = exit time, STEPMIN = last step size, ATOL/RTOL are set as INTEGRATE_x sets them, and an
 * unsuccessful integration prints the reference's message.  The ISO_C_BINDING shim in shim/ passes the COMMON block. */
int mistra_chem_integrate_common(int mech, void* gdata, double* tin, double* tout);

/* The same without the messages, for a caller that prints them itself as the reference does (the Fortran shim writes
 * ros_ErrorMsg_x's and INTEGRATE_x's lines to unit 6, gas.f:1474-1509, 764-767): *ierr = IERR of Rosenbrock_x (1 = success),
 * *t_err / *h_err = T and H when the integrator returned (the two numbers of ros_ErrorMsg_x's last line), *nsng = how
 * often the decomposition met a zero pivot (the reference prints one warning per occurrence, gas.f:1456).  Any of the four
 * may be NULL. */
int mistra_chem_integrate_common_status(int mech, void* gdata, double* tin, double* tout, int32_t* ierr, double* t_err,
                                        double* h_err, int32_t* nsng);

/* The rows of the zero pivots a cell met: what KppDecomp_x returns in IER — the first row k whose diagonal is exactly zero when the
 * elimination reaches it (gas.f:6157) — and ros_PrepareMatrix_x prints once per failed decomposition ("Warning: LU Decomposition
 * returned ising = k", gas.f:1456) before it halves H.  rows8[0 .. min(Nsng, 8)-1] = row numbers (1-based, = species numbers) in
 * order of occurrence for cell `cell` (0-based index into the batch) of the LAST host-buffer integration of this mechanism
 * (mistra_chem_integrate[_ex], mistra_chem_integrate_common[_status]); entries past Nsng are undefined.  The rows stay on the
 * device until asked for: call it only for cells whose statistics report Nsng > 0 (the Fortran shim does, to print the line). */
int mistra_chem_singular_rows(int mech, int cell, int32_t* rows8);

/* Update_RCONST_x for ncell cells (gas.f:275 | aer.f:304 | tot.f:1040; called by x_drive right before INTEGRATE_x, gas.f:172):
 * rconst[cell][NREACT] from env[cell][mistra_chem_rates_env_size(mech)], the per-cell inputs the generated routine and its
 * rate laws (kpp.f90:7127-8601) read from COMMON /cb_1/, /kpp_rate_x/, /ph_r_x/ and C.  Sizes: gas 74 doubles, aer 330,
 * tot 544 (instead of the 331 / 807 / 1627 rate constants); gas e.g.: aircc, te, h2oppm, pk | conv1, xhal, xiod, xhet1,
 * xhet2 | ycwd(1:2) | ph_rat(1:47) | FIX(1:3) | what fdhetg reads.  The layout of each mechanism is listed by name in
 * mistra_amd/mech/<mech>.rates_env.json (made by tools/extract_rates.py together with the rate table <mech>.rates the
 * library loads); Fortran array names there index as the reference declares them (kpp.f90:7140-7180).  Host buffers /
 * device buffers on a HIP stream.  The lazy initialisation of the Fortran-facing entry points applies to the host-buffer
 * form. */
int mistra_chem_rates_env_size(int mech);
int mistra_chem_update_rconst(int mech, int ncell, const double* env, double* rconst);
int mistra_chem_update_rconst_device(int mech, int ncell, const double* d_env, double* d_rconst, void* hip_stream);

/* ---- The hand-over halves of the per-layer drivers on the device: what gas_drive / aer_drive / tot_drive
 * (gas.f:60-217 | aer.f:59-246 | tot.f:59-982, with aer_mk.dat / aer_km.dat) do around Update_RCONST_x + INTEGRATE_x, for a batch of
 * layers ("cells") whose data stay in HBM.  Per layer k, cell-major:
 *   s1 [j1], s3 [j5]          s1(1:j1,k), s3(1:j5,k) of module gas_common (non-radical / radical gases)
 *   sl1 [nkc][j2]             sl1(1:j2,1:nkc,k) of COMMON /blck17/ (the layer's Fortran slab as it stands in memory), j2 = 121, nkc = 4
 *   sion1 [nkc][j6]           sion1(1:j6,1:nkc,k), j6 = 55
 *   scal [6]                  air, h2o, cvv1..cvv4 (the drivers' arguments; gas reads the first two, aer the first four)
 *   var [NVAR], fix [NFIX]    C = VAR | FIX of COMMON /GDATA_x/
 *   bg [NREACT][2]            bg(1:2,1:NREACT,kl) of COMMON /budg/ (bud_x.f): [..][0] instantaneous rate, [..][1] += dt * rate
 *   bgs [122][2]              bgs(1:2,1:122,k) of COMMON /budgs/ (bud_s_x.f); slots the mechanism does not set are left alone
 * Everything is index work and plain products in the reference's operation order: bit-identical results.
 *
 * mistra_chem_set_species_maps: the model's species index maps (module gas_common, built once by mk_interface, utils.f90:82-140, from
 * the user's species lists) for one mechanism: gas_m2k [j1][2] = gas_m2k_x(1:2,j) (C index, s1 index), gas_k2m [j1] = gas_k2m_x(j)
 * (C index of s1(j)), rad_* likewise for s3; 1-based as the Fortran holds them.  Host arrays; call once after init.  Restriction: every C index
 * must be a VARIABLE species (1..NVAR) — match_mk_indexes searches all NSPEC names, so a user list naming a fixed species (O2, N2, H2O ...) is
 * legal in the reference; here it is refused with "species map entry out of range" (no shipped gas_species / gas_radical list names one).
 * mistra_chem_drive_dims: j2, j6, nkc (global_params.f90:96-103) and the slot count of bgs.  Any pointer may be NULL. */
int mistra_chem_set_species_maps(int mech, int j1, const int32_t* gas_m2k, const int32_t* gas_k2m, int j5,
                                 const int32_t* rad_m2k, const int32_t* rad_k2m);
int mistra_chem_drive_dims(int mech, int* j2, int* j6, int* nkc, int* nbgs);

/* x_drive up to (not including) Update_RCONST_x (gas.f:132-167 | aer.f:152-214 | tot.f:212-599): C <- s1, s3 through the maps, FIX
 * from air / h2o / cvv, the liquid-phase species from sl1 / sion1 — which aer_drive and tot_drive first clamp to >= 0 IN PLACE
 * (tot.f:226-227), so d_sl1 / d_sion1 are in/out.  Entries of VAR that the driver does not set (KPP's dummy products) keep what
 * d_var holds, as COMMON /GDATA_x/ does in the reference.  Asynchronous on hip_stream. */
int mistra_chem_pack_device(int mech, int ncell, const double* d_s1, const double* d_s3, double* d_sl1, double* d_sion1,
                            const double* d_scal, double* d_var, double* d_fix, void* hip_stream);

/* The concentrations Update_RCONST_x reads (C(ind_Hplz), FIX(..) ...: the `c(i)` / `fix(i)` entries of the rate evaluator's input
 * vector, mistra_amd/mech/<mech>.rates_env.json) copied from the packed C into d_env [ncell][rates_env_size]; the other entries
 * (/cb_1/, /kpp_rate_x/, ph_rat: MISTRA_RATES_ENV_x of shim/mistra_kpp_rates.f90) are the caller's. */
int mistra_chem_rates_env_from_c_device(int mech, int ncell, const double* d_var, const double* d_fix, double* d_env, void* hip_stream);

/* bud_x (bud_g.f | bud_a.f | bud_t.f: every reaction's rate RCONST(i) * reactants, called by x_drive for the layers in il(1:nlev),
 * gas.f:179-184) and bud_s_x (bud_s_g.f | ..: the selected sulphur / DMS rates, every layer, gas.f:187) on the state the integration
 * left.  d_bg or d_bgs may be NULL (a caller passes to d_bg only the layers that are budget levels). */
int mistra_chem_budgets_device(int mech, int ncell, const double* d_var, const double* d_fix, const double* d_rconst, double dt,
                               double* d_bg, double* d_bgs, void* hip_stream);

/* The hand-over after the integration (gas.f:189-215 | aer.f:229-244 | tot.f:619-982): s1, s3 through the inverse maps, sl1 / sion1
 * from the liquid-phase species. */
int mistra_chem_unpack_device(int mech, int ncell, const double* d_var, double* d_s1, double* d_s3, double* d_sl1, double* d_sion1,
                              void* hip_stream);

/* One x_drive per layer for a batch of layers without anything but the model arrays crossing PCIe: pack -> env <- C -> Update_RCONST_x
 * -> INTEGRATE_x(tin, tin + dt) -> budgets -> hand-over, all on hip_stream.  d_env: the rate evaluator's input with the caller's
 * part filled in.  d_bg / d_bgs / d_texit_hexit may be NULL. */
int mistra_chem_drive_device(int mech, int ncell, double* d_s1, double* d_s3, double* d_sl1, double* d_sion1, const double* d_scal,
                             double* d_env, double* d_var, double* d_fix, double tin, double dt, int32_t* d_ierr, int32_t* d_stats,
                             double* d_texit_hexit, double* d_bg, double* d_bgs, void* hip_stream);

/* The same from the model's own arrays in HOST memory: what a Fortran caller hands over (shim/mistra_kpp_drive.f90: KPP_DRIVE_RUN) —
 * ONE call per mechanism and 10-s step for all layers that run it (kpp.f90:4454-4467 chooses the mechanism per layer), instead of one
 * x_drive per layer.  The arrays are the model's, dimensioned for n layers, layer k at its Fortran place:
 *   layer [nlayer]            the k of each layer of the batch, 1-based, in the order of the layer loop
 *   s1 [n][j1], s3 [n][j5]    s1(1:j1,1:n), s3(1:j5,1:n) of module gas_common (j1, j5: as given to mistra_chem_set_species_maps)
 *   sl1 [n][nkc][j2]          COMMON /blck17/ sl1(j2,nkc,n); sion1 [n][nkc][j6] likewise
 *   scal [nlayer][6], env [nlayer][rates_env_size]      per layer of the batch (not per k): air, h2o, cvv1..4 and the rate evaluator's input
 *                             with the caller's part filled in (MISTRA_RATES_ENV_x); the concentrations in it are refilled on the device
 *   bg [nlev][nrxn][2]        COMMON /budg/ bg(2,nrxn,nlev) (global_params.f90:110-115), or NULL; bg_level [nlayer]: kl (1-based) where layer
 *                             i is the budget level il(kl), else 0 (gas.f:179-184)
 *   bgs [n][122][2]           COMMON /budgs/ bgs(2,122,n), or NULL
 * in/out: s1, s3, sl1, sion1 (rows of the batch's layers), bg (rows of its levels), bgs.  Out, per layer of the batch, each may be NULL: ierr,
 * stats [8], t_h [3] as mistra_chem_integrate_ex returns them; c_packed [nlayer][NVAR+NFIX]: C = VAR | FIX as the pack half handed it to
 * INTEGRATE_x (diagnostics / tests).  KPP's dummy products, which the drivers never set, start from 0 in every layer (the reference carries
 * the previous layer's leftovers: INTEGRATION.md §4).  Synchronous; one pinned block up, the device chain on a private stream, one block down. */
int mistra_chem_drive(int mech, int nlayer, const int32_t* layer, int n, double* s1, double* s3, double* sl1, double* sion1, const double* scal,
                      const double* env, double tin, double dt, int32_t* ierr, int32_t* stats, double* t_h, double* bg, int nrxn,
                      const int32_t* bg_level, double* bgs, double* c_packed);
/* The same in two halves, so that the three mechanisms of one column step run SIDE BY SIDE on the device: kpp_driver gives every layer to one of
 * gas / aer / tot (kpp.f90:4454-4467), the three batches touch disjoint rows of the model arrays, and a column's 148 cells leave most of the GPU's 256 CUs
 * idle behind any one mechanism — a step then lasts as long as its slowest cell, not as the three mechanisms' slowest cells in a row (BTZ96: 14.4 -> 8.4 ms,
 * INTEGRATION.md §4d).  mistra_chem_drive_begin takes mistra_chem_drive's arguments, gathers the layers, and returns once the copies and kernels are
 * enqueued on the mechanism's private stream; mistra_chem_drive_end(mech) waits for them and scatters the results into the arrays given to begin.  Between
 * the two the caller must not touch those arrays' rows (nor ierr, stats, t_h, c_packed); one step per mechanism may be open at a time.
 * shim/mistra_kpp_drive.f90: KPP_DRIVE_RUN issues all mechanisms, then fetches them in mechanism order. */
int mistra_chem_drive_begin(int mech, int nlayer, const int32_t* layer, int n, double* s1, double* s3, double* sl1, double* sion1, const double* scal,
                      const double* env, double tin, double dt, int32_t* ierr, int32_t* stats, double* t_h, double* bg, int nrxn,
                      const int32_t* bg_level, double* bgs, double* c_packed);
int mistra_chem_drive_end(int mech);

/* ---- liq_parm, first slice: the gas <-> particle mass-transfer coefficients of fast_k_mt_a (mech = aer;
 * kpp.f90:2683-2947) and fast_k_mt_t (mech = tot; kpp.f90:2421-2676), called by liq_parm every 120 s (kpp.f90:617,637), for nlayer
 * layers at once.  Per layer, as the COMMON blocks hold them for that k:
 *   d_ff [nka][nkt]           ff(1:nkt,1:nka,k) of /cb52/: the 2-D particle spectrum (nkt = nka = 70)
 *   d_cw, d_cm [nkc]          cw(1:nkc,k), cm(1:nkc,k) of /blck12/ (liquid water content per chemical bin; cm is the activity switch)
 *   d_freep                   freep(k): mean free path (the routine's argument)
 *   d_alpha, d_vmean [NSPEC]  alpha(:,k), vmean(:,k) of /kpp_2aer/ | /kpp_2tot/ (accommodation coefficients, mean molecular speeds)
 *   d_xkmt [nkc][NSPEC]       xkmt(:,1:nkc,k) of /kpp_laer/ | /kpp_ltot/: in/out — written for the 50 exchanged species of the active
 *                             bins (cm > 0 and cw > 0), everything else left as it is, like the reference
 *   d_t, d_p                  t(k), p(k) of /cb53/: temperature and pressure (arguments of the terminal velocity vterm, str.f90:2793)
 *   d_vt [nkc]                vt(1:nkc,k) of /kpp_vt/: in/out — the LWC-weighted sedimentation velocity of every bin with cw > 0 ("computed
 *                             whatever LWC", kpp.f90:2421-2432: also for bins with cm = 0), read by SR sedl (str.f90:2704, 2751); bins with
 *                             cw <= 0 left as they are.  d_vt = NULL: not computed (then d_t, d_p may be NULL too)
 * and for the call: d_rq [nka][nkt] = rq(1:nkt,1:nka) of /cb50/ (particle radii, um), kw [nkw] (host; nkw must be nka = 70) and ka of /blck06/,
 * ifeed and nkc_l of module config.  The summation order is the reference's: bit-identical coefficients; vt likewise up to 10 um radius
 * (Stokes regime), to the last place of the device log / exp above (Beard's polynomial).  With d_vt the call does everything
 * fast_k_mt_a / fast_k_mt_t do: the Fortran call can be dropped.  Asynchronous on hip_stream; nothing is staged, kw travels with the launch. */
int mistra_chem_fast_k_mt_device(int mech, int nlayer, const double* d_ff, const double* d_rq, const int32_t* kw, int nkw, int ka, int ifeed,
                                 int nkc_l, const double* d_cw, const double* d_cm, const double* d_freep, const double* d_alpha,
                                 const double* d_vmean, double* d_xkmt, const double* d_t, const double* d_p, double* d_vt, void* hip_stream);

/* ---- liq_parm, second slice: the Henry constants of henry_a (mech = aer; kpp.f90:1914-2145) | henry_t (tot;
 * kpp.f90:1676-1907) and the forward / backward rate constants of the aqueous equilibria of equil_co_a (kpp.f90:3162-3363) | equil_co_t
 * (kpp.f90:2954-3155), which liq_parm calls every time step (kpp.f90:614-616, 634-636), for nlayer layers at once.  Per layer k:
 *   d_tt                      tt(k): temperature (/cb53/ t)
 *   d_henry [NSPEC]           henry(:,k) of /kpp_laer/ | /kpp_ltot/: written whole — the inverse dimensionless constant of the species
 *                             the routine lists, 0 for the others (NSPEC = NVAR + NFIX)
 *   d_conv2 [nkc]             conv2(1:nkc,k) of /blck13/ (1/(1000 cw); <= 0 = no liquid water in the bin)
 *   d_xgamma [nkc][j6]        xgamma(1:j6,1:nkc,k) of /kpp_mol/ (activity coefficients)
 *   d_xkef, d_xkeb [nkc][NSPEC]   xkef(:,1:nkc,k), xkeb(:,1:nkc,k) of /kpp_laer/ | /kpp_ltot/: in/out — a bin with conv2 <= 0 is zeroed, in
 *                             the others the listed species are written and the rest left as they are, like the reference; equil_co_a sets
 *                             bins 1..2 only, equil_co_t all four
 * The tables (mistra_amd/mech/<mech>.liq) are cut out of the reference source by tools/extract_liq.py; products are formed in the
 * reference's order, exp is the device library's (last-place differences against the host libm). */
int mistra_chem_henry_device(int mech, int nlayer, const double* d_tt, double* d_henry, void* hip_stream);
/* v_mean_a (tt,nmaxf) (kpp.f90:1472-1670) | v_mean_t (kpp.f90:1268-1465), which liq_parm calls every time step (kpp.f90:612, 632): the mean
 * molecular speeds vmean(:,k) = sqrt(tt(k)/M)*4.60138 of the species the routine lists, 0 for the others, for nlayer layers at once.
 *   d_tt [nlayer]             tt(k)
 *   d_vmean [nlayer][NSPEC]   vmean(:,k) of /kpp_2aer/ | /kpp_2tot/, written whole (NSPEC = NVAR + NFIX).  The reference zeroes the layers above
 *                             nmaxf as well (`vmean(:,:) = 0._dp`): a caller that replaces the routine does that once, they never change.
 * Table: mistra_amd/mech/<mech>.vmean, cut out of the reference source by tools/extract_vmean.py.  Quotient, square root and product round
 * once each as in the compiled reference: bit-identical (tests/test_gpu_liq.py). */
int mistra_chem_v_mean_device(int mech, int nlayer, const double* d_tt, double* d_vmean, void* hip_stream);
/* st_coeff_a (kpp.f90:857-1038) | st_coeff_t (kpp.f90:664-851), which liq_parm calls every time step (kpp.f90:614, 634): the accommodation
 * coefficients alpha(:,k) fast_k_mt_x reads, for nlayer layers at once.
 *   lp_joyce14bc, lp_buxmann15alph   the namelist switches of module config the routine branches on (alpha(NO3), alpha(N2O5) = a_n2o5(k,1),
 *                             kpp.f90:8377; alpha(ICl), alpha(IBr))
 *   d_env [nlayer][5]         per layer: t(k) (/cb53/), cw(1,k), cm(1,k) (/blck12/), sion1(13,1,k), sion1(14,1,k) (/blck17/) — the last four are
 *                             read by a_n2o5 only
 *   d_alpha [nlayer][NSPEC]   alpha(:,k) of /kpp_2aer/ | /kpp_2tot/, written whole: 0.1 for the species the routine does not list, min(1, .)
 *                             applied (NSPEC = NVAR + NFIX).  The reference computes layers 2..nf; layer 1 keeps the default 0.1.
 * Tables: mistra_amd/mech/<mech>.stcoeff (tools/extract_stcoeff.py: every assignment a postfix program, run by the evaluator of the rate
 * constants); the restated table is bit-exact against the running model on the CPU, the device to the last place of its exp. */
int mistra_chem_st_coeff_device(int mech, int nlayer, int lp_joyce14bc, int lp_buxmann15alph, const double* d_env, double* d_alpha, void* hip_stream);
int mistra_chem_equil_co_device(int mech, int nlayer, int nkc, int j6, const double* d_tt, const double* d_conv2, const double* d_xgamma,
                                double* d_xkef, double* d_xkeb, void* hip_stream);

/* The liq_parm kernels above on HOST buffers (same layouts, layer-major as the model holds them: every array of the reference has the
 * layer as its last dimension, so a run of layers kmin..kmax is handed over in place — ff(1,1,kmin), xkmt(1,1,kmin) ...): what the Fortran
 * shim calls (shim/mistra_kpp_liq.f90: FAST_K_MT_BATCH, HENRY_BATCH, V_MEAN_BATCH, ST_COEFF_BATCH, EQUIL_CO_BATCH, CW_RC_BATCH, DRY_RATES_BATCH; drop-ins with the reference's own signatures in
 * shim/mistra_kpp_model.f90).  Synchronous; primary device. */
int mistra_chem_fast_k_mt(int mech, int nlayer, const double* ff, const double* rq, const int32_t* kw, int nkw, int ka, int ifeed, int nkc_l,
                          const double* cw, const double* cm, const double* freep, const double* alpha, const double* vmean, double* xkmt,
                          const double* t, const double* p, double* vt);
int mistra_chem_henry(int mech, int nlayer, const double* tt, double* henry);
int mistra_chem_v_mean(int mech, int nlayer, const double* tt, double* vmean);
int mistra_chem_st_coeff(int mech, int nlayer, int lp_joyce14bc, int lp_buxmann15alph, const double* env, double* alpha);
/* cw_rc (nmaxf) (kpp.f90:2152-2414; liq_parm calls it every time step, kpp.f90:609) and, with dry != 0, dry_cw_rc (nmax) (kpp.f90:4580-4690): liquid water
 * content, mean radius, and — cw_rc — the water mass and the chemistry switch conv2 of the particle bins of nlayer layers, as moments of the
 * two-dimensional particle spectrum summed in the reference's own order (bit-identical: tests/test_gpu_liq.py).  Mechanism-independent.  Host buffers,
 * layer-major as the model holds them:
 *   ff [nlayer][nka][nkt]     ff(1:nkt,1:nka,k) of /cb52/;  rq [nka][nkt], e [nkt]: /cb50/;  kw [nka], ka: /blck06/;  ifeed: module config
 *   feu [nlayer]              relative humidity feu(k) of /cb54/;  cloud [nlayer][4]: cloud(1:nkc,k) of /kpp_l1/ as 0 | 1 (the hysteresis of bins 1, 2)
 *   crys4 [4]                 xcryssulf, xcrysss, xdelisulf, xdeliss of /kpp_crys/
 *   rc, cw, cm, conv2 [nlayer][4]   rc(:,k) of /blck11/, cw(:,k), cm(:,k) of /blck12/, conv2(:,k) of /blck13/;  with dry: rc, cw [nlayer][2] = rcd(:,k),
 *                             cwd(:,k) of the dry-aerosol common blocks, and e, feu, cloud, crys4, cm, conv2, below are not touched (may be NULL)
 *   below [nlayer]            1 where feu(k) < min(xcryssulf, xcrysss): the reference prints `k, feu(k), ' below both crystal. points'` for those
 *                             layers up to kinv (the Fortran drop-in does, shim/mistra_kpp_model.f90); may be NULL
 * cw_rc computes layers 2..nmaxf, dry_cw_rc nf+1..nmax: the caller hands over that run of layers. */
/* The host-buffer entries below gather their inputs into a pinned arena, send it up in one stream, and hand the outputs back the same way.  A caller whose
 * arrays never move — the model's COMMON blocks: ff of /cb52/ is 5.9 MB, /kpp_ltot/ 4.4 MB — registers them ONCE with mistra_chem_pin_host; from then on a
 * block that lies inside a registered range is copied straight from / to the caller's memory at the link's rate (no staging copy).  The range must stay
 * mapped until mistra_chem_unpin_host or mistra_chem_finalize; ranges must not overlap.  Not for memory the caller may free (the reference has no such
 * call: its arrays are static; shim/mistra_kpp_model.f90: LIQ_PIN_ONCE). */
int mistra_chem_pin_host(void* p, size_t bytes);
int mistra_chem_unpin_host(void* p);
/* dry_rates_g (tt,freep,nmax) (kpp.f90:4697-4853; gas != 0) | dry_rates_a (freep,nmaxf) (:4860-5073) | dry_rates_t (freep,nmaxf) (:5079-5198), called by
 * liq_parm every time step (kpp.f90:651-653): the mass-transfer coefficients of HNO3, N2O5, NH3, H2SO4 — the routines' idr list, in that order — onto the
 * dry aerosol of bins 1 and 2, and the equilibrium constant of HNO3.  The four species sit at mechanism-specific places of the model's NSPEC-wide arrays:
 * the caller (shim/mistra_kpp_model.f90: DRY_RATES_HIP_g/_a/_t) gathers vmean and scatters the results, so these arrays are compact.  Host buffers:
 *   tt, freep [nlayer]        temperature (/cb53/ t; dry_rates_g: its argument tt) and mean free path of the layers k = 2..nmax
 *   rcd [nlayer][2]           rcd(1:2,k) of /blck11/
 *   vmean4 [nlayer][4]        vmean(idr(l),k) of /kpp_2aer/ | /kpp_2tot/ (aer, tot; NULL for gas: dry_rates_g forms its own)
 *   xkmtd [nlayer][2][4]      out: xkmtd(idr(l),kc,k);   xeq [nlayer]: out: xeq(ind_HNO3,k)
 *   henry4 [nlayer][4]        gas only, in/out: henry(idr(l),k) of /kpp_dryg/ (HNO3 is set, then every positive entry becomes 1/(henry*FCT), as the reference does) */
int mistra_chem_dry_rates(int gas, int nlayer, const double* tt, const double* freep, const double* rcd, const double* vmean4, double* xkmtd, double* xeq,
                          double* henry4);
int mistra_chem_cw_rc(int nlayer, int nkt, int nka, int dry, const double* ff, const double* rq, const double* e, const int32_t* kw, int ka, int ifeed,
                      const double* feu, const int32_t* cloud, const double* crys4, double* rc, double* cw, double* cm, double* conv2, int32_t* below);
int mistra_chem_equil_co(int mech, int nlayer, int nkc, int j6, const double* tt, const double* conv2, const double* xgamma, double* xkef,
                         double* xkeb);

/* Diagnostics for the phase-level parity tests: integrates the cells like mistra_chem_integrate (results discarded) with the
 * kernel variant that writes out, per cell, the intermediate results of the FIRST attempt of the first Rosenbrock step
 * (gas.f:1201-1262): dump[cell][5*NVAR + 2*LU_NONZERO + 2] = Fcn0 (Fun_x) | Ghimj as ros_PrepareMatrix_x builds it | Ghimj
 * after KppDecomp_x in the kernel's form (multipliers in L; the rows of the solves' tail chain row-scaled by 1/U(k,k)) |
 * 1/U(k,k) | K(1) | K(2) | K(3) (KppSolve_x results of the three stages) | Err (ros_ErrorNorm_x) | H. */
int mistra_chem_debug_first_step(int mech, int ncell, const double* var_in, const double* fix, const double* rconst,
                                 double tin, double tout, double* dump);

/* Test hook for the one exit of RosenbrockIntegrator_x that INTEGRATE_x's fixed options put out of a test's reach: IERR = -6, "No of steps
 * exceeds maximum bound" (gas.f:1199-1202), taken when Nstp > Max_no_steps = IPAR(3), which INTEGRATE_x leaves at its default of 100000
 * (gas.f:729-732, 845-846, 1042).  Sets that bound for every later integrate call of the process; max_steps <= 0 restores 100000.  The
 * compiled reference is driven the same way (Rosenbrock_x called with IPAR(3) set) in tests/test_oracle.py. */
int mistra_chem_debug_set_max_steps(int max_steps);

/* Text of the last error on this thread ("" if none). */
const char* mistra_chem_last_error(void);

/* One line describing the schedule built for a mechanism (rounds, slots); for logs. */
const char* mistra_chem_describe(int mech);

#ifdef __cplusplus
}
#endif
#endif /* MISTRA_CHEM_H */
