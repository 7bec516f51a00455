! Fortran side of liq_parm's table-driven kernels on the GPU (SURVEY.md §8 f3; include/mistra_chem.h: mistra_chem_fast_k_mt, _henry, _v_mean, _st_coeff, _equil_co, _cw_rc, _dry_rates):
!   FAST_K_MT_BATCH   fast_k_mt_a (kpp.f90:2683-2947) | fast_k_mt_t (kpp.f90:2421-2676): xkmt AND the sedimentation velocity vt, every 120 s
!   HENRY_BATCH       henry_a (kpp.f90:1914-2145)     | henry_t (kpp.f90:1676-1907): the inverse dimensionless Henry constants, every step
!   V_MEAN_BATCH      v_mean_a (kpp.f90:1472-1670)    | v_mean_t (kpp.f90:1268-1465): the mean molecular speeds, every step
!   ST_COEFF_BATCH    st_coeff_a (kpp.f90:857-1038)   | st_coeff_t (kpp.f90:664-851): the accommodation coefficients alpha, every step
!   CW_RC_BATCH       cw_rc (kpp.f90:2152-2414) | dry_cw_rc (kpp.f90:4580-4690): liquid water, mean radius, water mass and chemistry switch of the particle bins
!   DRY_RATES_BATCH   dry_rates_g (kpp.f90:4697-4853) | dry_rates_a (:4860-5073) | dry_rates_t (:5079-5198): uptake of four species on the dry aerosol, every step
!   EQUIL_CO_BATCH    equil_co_a (kpp.f90:3162-3363)  | equil_co_t (kpp.f90:2954-3155): forward / backward equilibrium rate constants, every step
! for a run of consecutive layers.  Every array of the reference has the layer as its LAST dimension, so the caller hands over the model
! arrays in place, starting at the first layer of the run: ff(1,1,kmin), xkmt(1,1,kmin), cw(1,kmin), freep(kmin) ... (drop-ins with the
! reference's own argument lists and COMMON blocks: shim/mistra_kpp_model.f90).  mech: 2 = aer, 3 = tot (1 = gas has no such routine).
module mistra_kpp_liq
  use iso_c_binding
  use mistra_chem_c_api, only: mistra_chem_fail
  implicit none
  interface
     function mistra_chem_fast_k_mt(mech, nlayer, ff, rq, kw, nkw, ka, ifeed, nkc_l, cw, cm, freep, alpha, vmean, xkmt, t, p, vt) &
          bind(C, name="mistra_chem_fast_k_mt") result(rc)
       import :: c_int, c_int32_t, c_double
       integer(c_int), value :: mech, nlayer, nkw, ka, ifeed, nkc_l
       integer(c_int32_t), intent(in) :: kw(*)
       real(c_double), intent(in) :: ff(*), rq(*), cw(*), cm(*), freep(*), alpha(*), vmean(*), t(*), p(*)
       real(c_double) :: xkmt(*), vt(*)
       integer(c_int) :: rc
     end function mistra_chem_fast_k_mt
     function mistra_chem_henry(mech, nlayer, tt, henry) bind(C, name="mistra_chem_henry") result(rc)
       import :: c_int, c_double
       integer(c_int), value :: mech, nlayer
       real(c_double), intent(in) :: tt(*)
       real(c_double) :: henry(*)
       integer(c_int) :: rc
     end function mistra_chem_henry
     function mistra_chem_v_mean(mech, nlayer, tt, vmean) bind(C, name="mistra_chem_v_mean") result(rc)
       import :: c_int, c_double
       integer(c_int), value :: mech, nlayer
       real(c_double), intent(in) :: tt(*)
       real(c_double) :: vmean(*)
       integer(c_int) :: rc
     end function mistra_chem_v_mean
     function mistra_chem_st_coeff(mech, nlayer, lp_joyce14bc, lp_buxmann15alph, env, alpha) bind(C, name="mistra_chem_st_coeff") result(rc)
       import :: c_int, c_double
       integer(c_int), value :: mech, nlayer, lp_joyce14bc, lp_buxmann15alph
       real(c_double), intent(in) :: env(*)
       real(c_double) :: alpha(*)
       integer(c_int) :: rc
     end function mistra_chem_st_coeff
     function mistra_chem_cw_rc(nlayer, nkt, nka, dry, ff, rq, e, kw, ka, ifeed, feu, cloud, crys4, rc, cw, cm, conv2, below) &
          bind(C, name="mistra_chem_cw_rc") result(ret)
       import :: c_int, c_int32_t, c_double
       integer(c_int), value :: nlayer, nkt, nka, dry, ka, ifeed
       real(c_double), intent(in) :: ff(*), rq(*), e(*), feu(*), crys4(4)
       integer(c_int32_t), intent(in) :: kw(*), cloud(*)
       real(c_double) :: rc(*), cw(*), cm(*), conv2(*)
       integer(c_int32_t) :: below(*)
       integer(c_int) :: ret
     end function mistra_chem_cw_rc
     function mistra_chem_dry_rates(gas, nlayer, tt, freep, rcd, vmean4, xkmtd, xeq, henry4) bind(C, name="mistra_chem_dry_rates") result(rc)
       import :: c_int, c_double
       integer(c_int), value :: gas, nlayer
       real(c_double), intent(in) :: tt(*), freep(*), rcd(*), vmean4(*)
       real(c_double) :: xkmtd(*), xeq(*), henry4(*)
       integer(c_int) :: rc
     end function mistra_chem_dry_rates
     function mistra_chem_equil_co(mech, nlayer, nkc, j6, tt, conv2, xgamma, xkef, xkeb) bind(C, name="mistra_chem_equil_co") result(rc)
       import :: c_int, c_double
       integer(c_int), value :: mech, nlayer, nkc, j6
       real(c_double), intent(in) :: tt(*), conv2(*), xgamma(*)
       real(c_double) :: xkef(*), xkeb(*)
       integer(c_int) :: rc
     end function mistra_chem_equil_co
     function mistra_chem_pin_host(p, bytes) bind(C, name="mistra_chem_pin_host") result(rc)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: p
       integer(c_size_t), value :: bytes
       integer(c_int) :: rc
     end function mistra_chem_pin_host
  end interface
contains
  ! registers n doubles starting at a (a model array that never moves) for direct transfers; a refusal is not an error: the call then stages as before
  subroutine PIN_HOST(a, n)
    real(c_double), target, intent(in) :: a(*)
    integer, intent(in) :: n
    if (mistra_chem_pin_host(c_loc(a), int(n, c_size_t) * 8_c_size_t) /= 0) print *, 'mistra_kpp_liq: model array not registered, transfers are staged'
  end subroutine PIN_HOST

  ! ff(nkt,nka,nlayer), rq(nkt,nka), kw(nka), cw / cm(nkc,nlayer), freep(nlayer), alpha / vmean(NSPEC,nlayer), xkmt(NSPEC,nkc,nlayer) in/out,
  ! t / p(nlayer), vt(nkc,nlayer) in/out
  subroutine FAST_K_MT_BATCH(mech, nlayer, ff, rq, nka, kw, ka, ifeed, nkc_l, cw, cm, freep, alpha, vmean, xkmt, t, p, vt)
    integer, intent(in) :: mech, nlayer, nka, kw(nka), ka, ifeed, nkc_l
    real(c_double), intent(in) :: ff(*), rq(*), cw(*), cm(*), freep(*), alpha(*), vmean(*), t(*), p(*)
    real(c_double) :: xkmt(*), vt(*)
    if (nlayer <= 0) return
    if (mistra_chem_fast_k_mt(int(mech - 1, c_int), int(nlayer, c_int), ff, rq, int(kw, c_int32_t), int(nka, c_int), int(ka, c_int), int(ifeed, c_int), &
                              int(nkc_l, c_int), cw, cm, freep, alpha, vmean, xkmt, t, p, vt) /= 0) call mistra_chem_fail('FAST_K_MT_BATCH')
  end subroutine FAST_K_MT_BATCH

  ! tt(nlayer) -> henry(NSPEC,nlayer), written whole
  subroutine HENRY_BATCH(mech, nlayer, tt, henry)
    integer, intent(in) :: mech, nlayer
    real(c_double), intent(in) :: tt(*)
    real(c_double) :: henry(*)
    if (nlayer <= 0) return
    if (mistra_chem_henry(int(mech - 1, c_int), int(nlayer, c_int), tt, henry) /= 0) call mistra_chem_fail('HENRY_BATCH')
  end subroutine HENRY_BATCH

  ! tt(nlayer) -> vmean(NSPEC,nlayer), written whole
  subroutine V_MEAN_BATCH(mech, nlayer, tt, vmean)
    integer, intent(in) :: mech, nlayer
    real(c_double), intent(in) :: tt(*)
    real(c_double) :: vmean(*)
    if (nlayer <= 0) return
    if (mistra_chem_v_mean(int(mech - 1, c_int), int(nlayer, c_int), tt, vmean) /= 0) call mistra_chem_fail('V_MEAN_BATCH')
  end subroutine V_MEAN_BATCH

  ! env(5,nlayer) = t(k), cw(1,k), cm(1,k), sion1(13,1,k), sion1(14,1,k) -> alpha(NSPEC,nlayer), written whole; the two namelist switches of module config
  subroutine ST_COEFF_BATCH(mech, nlayer, lpJoyce14bc, lpBuxmann15alph, env, alpha)
    integer, intent(in) :: mech, nlayer
    logical, intent(in) :: lpJoyce14bc, lpBuxmann15alph
    real(c_double), intent(in) :: env(*)
    real(c_double) :: alpha(*)
    if (nlayer <= 0) return
    if (mistra_chem_st_coeff(int(mech - 1, c_int), int(nlayer, c_int), merge(1_c_int, 0_c_int, lpJoyce14bc), merge(1_c_int, 0_c_int, lpBuxmann15alph), &
                             env, alpha) /= 0) call mistra_chem_fail('ST_COEFF_BATCH')
  end subroutine ST_COEFF_BATCH

  ! ff(nkt,nka,nlayer), rq(nkt,nka), e(nkt), kw(nka), feu(nlayer), cloud(4,nlayer) (0 | 1), crys4 = xcryssulf, xcrysss, xdelisulf, xdeliss
  ! -> rc, cw, cm, conv2 (4,nlayer), below(nlayer); dry: rc, cw (2,nlayer) = rcd, cwd, the rest untouched
  subroutine CW_RC_BATCH(nlayer, nkt, nka, dry, ff, rq, e, kw, ka, ifeed, feu, cloud, crys4, rc, cw, cm, conv2, below)
    integer, intent(in) :: nlayer, nkt, nka, kw(nka), ka, ifeed, cloud(*)
    logical, intent(in) :: dry
    real(c_double), intent(in) :: ff(*), rq(*), e(*), feu(*), crys4(4)
    real(c_double) :: rc(*), cw(*), cm(*), conv2(*)
    integer :: below(*)
    integer(c_int32_t), allocatable :: cl(:), bl(:)
    if (nlayer <= 0) return
    allocate (cl(4 * nlayer), bl(nlayer))
    if (.not. dry) cl = int(cloud(1:4 * nlayer), c_int32_t)
    if (mistra_chem_cw_rc(int(nlayer, c_int), int(nkt, c_int), int(nka, c_int), merge(1_c_int, 0_c_int, dry), ff, rq, e, int(kw, c_int32_t), int(ka, c_int), &
                          int(ifeed, c_int), feu, cl, crys4, rc, cw, cm, conv2, bl) /= 0) call mistra_chem_fail('CW_RC_BATCH')
    if (.not. dry) below(1:nlayer) = bl
  end subroutine CW_RC_BATCH

  ! tt, freep(nlayer), rcd(2,nlayer), vmean4(4,nlayer) (aer, tot) -> xkmtd(4,2,nlayer), xeq(nlayer); gas: henry4(4,nlayer) in/out.  Species order: the
  ! routines' idr list HNO3, N2O5, NH3, H2SO4
  subroutine DRY_RATES_BATCH(gas, nlayer, tt, freep, rcd, vmean4, xkmtd, xeq, henry4)
    logical, intent(in) :: gas
    integer, intent(in) :: nlayer
    real(c_double), intent(in) :: tt(*), freep(*), rcd(*), vmean4(*)
    real(c_double) :: xkmtd(*), xeq(*), henry4(*)
    if (nlayer <= 0) return
    if (mistra_chem_dry_rates(merge(1_c_int, 0_c_int, gas), int(nlayer, c_int), tt, freep, rcd, vmean4, xkmtd, xeq, henry4) /= 0) &
         call mistra_chem_fail('DRY_RATES_BATCH')
  end subroutine DRY_RATES_BATCH

  ! tt(nlayer), conv2(nkc,nlayer), xgamma(j6,nkc,nlayer) -> xkef, xkeb(NSPEC,nkc,nlayer) in/out
  subroutine EQUIL_CO_BATCH(mech, nlayer, nkc, j6, tt, conv2, xgamma, xkef, xkeb)
    integer, intent(in) :: mech, nlayer, nkc, j6
    real(c_double), intent(in) :: tt(*), conv2(*), xgamma(*)
    real(c_double) :: xkef(*), xkeb(*)
    if (nlayer <= 0) return
    if (mistra_chem_equil_co(int(mech - 1, c_int), int(nlayer, c_int), int(nkc, c_int), int(j6, c_int), tt, conv2, xgamma, xkef, xkeb) /= 0) &
         call mistra_chem_fail('EQUIL_CO_BATCH')
  end subroutine EQUIL_CO_BATCH
end module mistra_kpp_liq
