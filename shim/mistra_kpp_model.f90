! The ONE shim file that is compiled against the model's own modules and COMMON declarations (put it behind common_modules.f90 and
! mistra_kpp_drive.f90 in src/Makefile): KPP_DRIVE_RUN is what the patched kpp_driver calls behind its layer loop (shim/kpp_drive.patch).
! It names the model arrays the per-layer drivers read and write — module gas_common (common_modules.f90:77-133: s1, s3 and the species
! index maps mk_interface builds, utils.f90:82-140), COMMON /blck17/ (gas.f:101), /budg/ (gas.f:104), /budgs/ (bud_s_g.f:63) — and hands
! them, as they stand in memory, to kpp_drive_run_arrays (shim/mistra_kpp_drive.f90), i.e. to ONE mistra_chem_drive call per mechanism.
subroutine KPP_DRIVE_RUN(tkpp, dt_ch)
  USE gas_common, ONLY : j1, j5, s1, s3, gas_m2k_g, gas_k2m_g, rad_m2k_g, rad_k2m_g, gas_m2k_a, gas_k2m_a, rad_m2k_a, rad_k2m_a, &
       gas_m2k_t, gas_k2m_t, rad_m2k_t, rad_k2m_t
  USE global_params, ONLY : j2, j6, n, nkc, nlev, nrxn
  USE mistra_kpp_drive, ONLY : kpp_drive_maps, kpp_drive_run_arrays
  implicit none
  double precision, intent(in) :: tkpp, dt_ch
  common /blck17/ sl1(j2,nkc,n), sion1(j6,nkc,n)
  double precision :: sl1, sion1
  common /budg/ bg(2,nrxn,nlev), il(nlev)
  double precision :: bg
  integer :: il
  common /budgs/ bgs(2,122,n)
  double precision :: bgs
  logical, save :: first = .true.
  if (first) then
     call kpp_drive_maps(1, j1, gas_m2k_g, gas_k2m_g, j5, rad_m2k_g, rad_k2m_g)
     call kpp_drive_maps(2, j1, gas_m2k_a, gas_k2m_a, j5, rad_m2k_a, rad_k2m_a)
     call kpp_drive_maps(3, j1, gas_m2k_t, gas_k2m_t, j5, rad_m2k_t, rad_k2m_t)
     first = .false.
  end if
  call kpp_drive_run_arrays(tkpp, dt_ch, n, s1, s3, sl1, sion1, nrxn, nlev, il, bg, bgs)
end subroutine KPP_DRIVE_RUN

! ---- liq_parm's kernels (SURVEY.md §8 f3) with the reference's own argument lists: rename these to fast_k_mt_t / henry_t / equil_co_t (and
!      the _a ones) in place of the reference's routines, or call them from liq_parm (kpp.f90:614-637).  They name the model's COMMON blocks as
!      the reference routines do (kpp.f90:2483-2526 | 2745-2788; 1717-1722; 3008-3016) and hand the layers nmin..nmax over in place.
subroutine FAST_K_MT_HIP_t(freep, box, n_bl)      ! fast_k_mt_t (freep,box,n_bl), kpp.f90:2421
  USE config, ONLY : ifeed
  USE global_params, ONLY : nf, n, nka, nkt, nkc
  USE mistra_kpp_liq, ONLY : FAST_K_MT_BATCH
  implicit none
  double precision, intent(in) :: freep(n)
  logical, intent(in) :: box
  integer, intent(in) :: n_bl
  integer, parameter :: NSPEC = 424                ! tot_Parameters.h
  integer :: kw, ka, nar, nmin, nmax
  double precision :: cw, cm, enw, ew, rn, rw, en, e, dew, rq, ff, fsum, theta, thetl, t, talt, p, rho, alpha, vmean, henry, xkmt, xkef, xkeb, vt, vd, vdm
  common /blck06/ kw(nka), ka
  common /blck12/ cw(nkc,n), cm(nkc,n)
  common /cb50/ enw(nka), ew(nkt), rn(nka), rw(nkt,nka), en(nka), e(nkt), dew(nkt), rq(nkt,nka)
  common /cb52/ ff(nkt,nka,n), fsum(n), nar(n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_2tot/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  common /kpp_ltot/ henry(NSPEC,nf), xkmt(NSPEC,nkc,nf), xkef(NSPEC,nkc,nf), xkeb(NSPEC,nkc,nf)
  common /kpp_vt/ vt(nkc,nf), vd(nkt,nka), vdm(nkc)
  nmin = 2; nmax = nf                              ! kpp.f90:2541-2547
  if (box) then
     nmin = n_bl; nmax = n_bl
  end if
  call FAST_K_MT_BATCH(3, nmax - nmin + 1, ff(1,1,nmin), rq, nka, kw, ka, ifeed, nkc, cw(1,nmin), cm(1,nmin), freep(nmin), alpha(1,nmin), vmean(1,nmin), &
                       xkmt(1,1,nmin), t(nmin), p(nmin), vt(1,nmin))
end subroutine FAST_K_MT_HIP_t

subroutine FAST_K_MT_HIP_a(freep, box, n_bl)      ! fast_k_mt_a (freep,box,n_bl), kpp.f90:2683
  USE config, ONLY : ifeed, nkc_l
  USE global_params, ONLY : nf, n, nka, nkt, nkc
  USE mistra_kpp_liq, ONLY : FAST_K_MT_BATCH
  implicit none
  double precision, intent(in) :: freep(n)
  logical, intent(in) :: box
  integer, intent(in) :: n_bl
  integer, parameter :: NSPEC = 262                ! aer_Parameters.h
  integer :: kw, ka, nar, nmin, nmax
  double precision :: cw, cm, enw, ew, rn, rw, en, e, dew, rq, ff, fsum, theta, thetl, t, talt, p, rho, alpha, vmean, henry, xkmt, xkef, xkeb, vt, vd, vdm
  common /blck06/ kw(nka), ka
  common /blck12/ cw(nkc,n), cm(nkc,n)
  common /cb50/ enw(nka), ew(nkt), rn(nka), rw(nkt,nka), en(nka), e(nkt), dew(nkt), rq(nkt,nka)
  common /cb52/ ff(nkt,nka,n), fsum(n), nar(n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_2aer/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  common /kpp_laer/ henry(NSPEC,nf), xkmt(NSPEC,nkc,nf), xkef(NSPEC,nkc,nf), xkeb(NSPEC,nkc,nf)
  common /kpp_vt/ vt(nkc,nf), vd(nkt,nka), vdm(nkc)
  nmin = 2; nmax = nf
  if (box) then
     nmin = n_bl; nmax = n_bl
  end if
  call FAST_K_MT_BATCH(2, nmax - nmin + 1, ff(1,1,nmin), rq, nka, kw, ka, ifeed, nkc_l, cw(1,nmin), cm(1,nmin), freep(nmin), alpha(1,nmin), vmean(1,nmin), &
                       xkmt(1,1,nmin), t(nmin), p(nmin), vt(1,nmin))
end subroutine FAST_K_MT_HIP_a

subroutine HENRY_HIP_t(tt, nmaxf)                  ! henry_t (tt,nmaxf), kpp.f90:1676: layers 1..nmaxf
  USE global_params, ONLY : nf, n, nkc
  USE mistra_kpp_liq, ONLY : HENRY_BATCH
  implicit none
  double precision, intent(in) :: tt(n)
  integer, intent(in) :: nmaxf
  integer, parameter :: NSPEC = 424
  double precision :: henry, xkmt, xkef, xkeb
  common /kpp_ltot/ henry(NSPEC,nf), xkmt(NSPEC,nkc,nf), xkef(NSPEC,nkc,nf), xkeb(NSPEC,nkc,nf)
  call HENRY_BATCH(3, nmaxf, tt, henry)
end subroutine HENRY_HIP_t

subroutine HENRY_HIP_a(tt, nmaxf)                  ! henry_a (tt,nmaxf), kpp.f90:1914
  USE global_params, ONLY : nf, n, nkc
  USE mistra_kpp_liq, ONLY : HENRY_BATCH
  implicit none
  double precision, intent(in) :: tt(n)
  integer, intent(in) :: nmaxf
  integer, parameter :: NSPEC = 262
  double precision :: henry, xkmt, xkef, xkeb
  common /kpp_laer/ henry(NSPEC,nf), xkmt(NSPEC,nkc,nf), xkef(NSPEC,nkc,nf), xkeb(NSPEC,nkc,nf)
  call HENRY_BATCH(2, nmaxf, tt, henry)
end subroutine HENRY_HIP_a

subroutine DRY_RATES_HIP_g(tt, freep, nmax)        ! dry_rates_g (tt,freep,nmax), kpp.f90:4697: layers 2..nmax
  USE global_params, ONLY : n, nkc
  USE mistra_kpp_liq, ONLY : DRY_RATES_BATCH
  implicit none
  include 'gas_Parameters.h'
  double precision, intent(in) :: tt(n), freep(n)
  integer, intent(in) :: nmax
  double precision :: rcd, xkmtd, henry, xeq, xk(4, 2, n), xq(n), h4(4, n), dum(1)
  integer :: idr(4), k, kc
  common /blck11/ rcd(nkc,n)
  common /kpp_dryg/ xkmtd(NSPEC,2,n), henry(NSPEC,n), xeq(NSPEC,n)
  if (nmax < 2) return
  idr = [ind_HNO3, ind_N2O5, ind_NH3, ind_H2SO4]
  do k = 2, nmax
     h4(:, k) = henry(idr, k)
  end do
  call DRY_RATES_BATCH(.true., nmax - 1, tt(2), freep(2), rcd(1:2, 2:nmax), dum, xk(1,1,2), xq(2), h4(1,2))
  do k = 2, nmax
     xeq(ind_HNO3, k) = xq(k)
     henry(idr, k) = h4(:, k)
     do kc = 1, 2
        xkmtd(idr, kc, k) = xk(:, kc, k)
     end do
  end do
end subroutine DRY_RATES_HIP_g

subroutine DRY_RATES_HIP_a(freep, nmaxf)          ! dry_rates_a (freep,nmaxf), kpp.f90:4860: layers 2..nmaxf
  USE global_params, ONLY : nf, n, nkc
  USE mistra_kpp_liq, ONLY : DRY_RATES_BATCH
  implicit none
  include 'aer_Parameters.h'
  double precision, intent(in) :: freep(n)
  integer, intent(in) :: nmaxf
  double precision :: rcd, xkmtd, xeq, alpha, vmean, theta, thetl, t, talt, p, rho, xk(4, 2, nf), xq(nf), v4(4, nf), dum(1)
  integer :: idr(4), k, kc
  common /blck11/ rcd(nkc,n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_2aer/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  common /kpp_drya/ xkmtd(NSPEC,2,nf), xeq(NSPEC,nf)
  if (nmaxf < 2) return
  idr = [ind_HNO3, ind_N2O5, ind_NH3, ind_H2SO4]
  do k = 2, nmaxf
     v4(:, k) = vmean(idr, k)
  end do
  call DRY_RATES_BATCH(.false., nmaxf - 1, t(2), freep(2), rcd(1:2, 2:nmaxf), v4(1,2), xk(1,1,2), xq(2), dum)
  do k = 2, nmaxf
     xeq(ind_HNO3, k) = xq(k)
     do kc = 1, 2
        xkmtd(idr, kc, k) = xk(:, kc, k)
     end do
  end do
end subroutine DRY_RATES_HIP_a

subroutine DRY_RATES_HIP_t(freep, nmaxf)          ! dry_rates_t (freep,nmaxf), kpp.f90:5079: layers 2..nmaxf
  USE global_params, ONLY : nf, n, nkc
  USE mistra_kpp_liq, ONLY : DRY_RATES_BATCH
  implicit none
  include 'tot_Parameters.h'
  double precision, intent(in) :: freep(n)
  integer, intent(in) :: nmaxf
  double precision :: rcd, xkmtd, xeq, alpha, vmean, theta, thetl, t, talt, p, rho, xk(4, 2, nf), xq(nf), v4(4, nf), dum(1)
  integer :: idr(4), k, kc
  common /blck11/ rcd(nkc,n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_2tot/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  common /kpp_dryt/ xkmtd(NSPEC,2,nf), xeq(NSPEC,nf)
  if (nmaxf < 2) return
  idr = [ind_HNO3, ind_N2O5, ind_NH3, ind_H2SO4]
  do k = 2, nmaxf
     v4(:, k) = vmean(idr, k)
  end do
  call DRY_RATES_BATCH(.false., nmaxf - 1, t(2), freep(2), rcd(1:2, 2:nmaxf), v4(1,2), xk(1,1,2), xq(2), dum)
  do k = 2, nmaxf
     xeq(ind_HNO3, k) = xq(k)
     do kc = 1, 2
        xkmtd(idr, kc, k) = xk(:, kc, k)
     end do
  end do
end subroutine DRY_RATES_HIP_t

subroutine LIQ_PIN_ONCE      ! the model arrays liq_parm's kernels move in bulk, registered for direct transfers on the first call (include/mistra_chem.h)
  USE global_params, ONLY : n, nf, nka, nkt, nkc
  USE mistra_kpp_liq, ONLY : PIN_HOST
  implicit none
  logical, save :: done = .false.
  integer, parameter :: NSPEC_a = 262, NSPEC_t = 424
  integer :: nar
  double precision :: ff, fsum, henry_a, xkmt_a, xkef_a, xkeb_a, henry_t, xkmt_t, xkef_t, xkeb_t
  common /cb52/ ff(nkt,nka,n), fsum(n), nar(n)
  common /kpp_laer/ henry_a(NSPEC_a,nf), xkmt_a(NSPEC_a,nkc,nf), xkef_a(NSPEC_a,nkc,nf), xkeb_a(NSPEC_a,nkc,nf)
  common /kpp_ltot/ henry_t(NSPEC_t,nf), xkmt_t(NSPEC_t,nkc,nf), xkef_t(NSPEC_t,nkc,nf), xkeb_t(NSPEC_t,nkc,nf)
  if (done) return
  done = .true.
  call PIN_HOST(ff, nkt * nka * n)
  call PIN_HOST(henry_a, NSPEC_a * nf * (1 + 3 * nkc))      ! the four arrays of the block are contiguous
  call PIN_HOST(henry_t, NSPEC_t * nf * (1 + 3 * nkc))
end subroutine LIQ_PIN_ONCE

subroutine CW_RC_HIP(nmaxf)                        ! cw_rc (nmaxf), kpp.f90:2152: layers 2..nmaxf
  USE config, ONLY : ifeed
  USE global_params, ONLY : n, nka, nkt, nkc
  USE mistra_kpp_liq, ONLY : CW_RC_BATCH
  implicit none
  integer, intent(in) :: nmaxf
  integer :: kw, ka, nar, kinv, k, cl(nkc, n), below(n)
  double precision :: rc, cw, cm, conv2, enw, ew, rn, rw, en, e, dew, rq, ff, fsum, xm1, xm2, feu, dfddt, xm1a, xcryssulf, xcrysss, xdelisulf, xdeliss
  logical :: cloud
  common /blck06/ kw(nka), ka
  common /blck11/ rc(nkc,n)
  common /blck12/ cw(nkc,n), cm(nkc,n)
  common /blck13/ conv2(nkc,n)
  common /cb50/ enw(nka), ew(nkt), rn(nka), rw(nkt,nka), en(nka), e(nkt), dew(nkt), rq(nkt,nka)
  common /cb52/ ff(nkt,nka,n), fsum(n), nar(n)
  common /cb54/ xm1(n), xm2(n), feu(n), dfddt(n), xm1a(n)
  common /kpp_l1/ cloud(nkc,n)
  common /kpp_crys/ xcryssulf, xcrysss, xdelisulf, xdeliss
  common /kinv_i/ kinv
  if (nmaxf < 2) return
  call LIQ_PIN_ONCE      ! (cw_rc is liq_parm's first kernel call, kpp.f90:609)
  cl = merge(1, 0, cloud)
  call CW_RC_BATCH(nmaxf - 1, nkt, nka, .false., ff(1,1,2), rq, e, kw, ka, ifeed, feu(2), cl(1,2), [xcryssulf, xcrysss, xdelisulf, xdeliss], &
                   rc(1,2), cw(1,2), cm(1,2), conv2(1,2), below(2))
  do k = 2, nmaxf                                 ! kpp.f90:2335
     if (below(k) /= 0 .and. k <= kinv) print *, k, feu(k), ' below both crystal. points'
  end do
end subroutine CW_RC_HIP

subroutine DRY_CW_RC_HIP(nmax)                     ! dry_cw_rc (nmax), kpp.f90:4580: layers nf+1..nmax, bins 1 and 2 of rc / cw
  USE config, ONLY : ifeed
  USE global_params, ONLY : nf, n, nka, nkt, nkc
  USE mistra_kpp_liq, ONLY : CW_RC_BATCH
  implicit none
  integer, intent(in) :: nmax
  integer :: kw, ka, nar, k, idum(1)
  double precision :: rc, cw, cm, enw, ew, rn, rw, en, e, dew, rq, ff, fsum, rcd(2, n), cwd(2, n), ddum(4)
  common /blck06/ kw(nka), ka
  common /blck11/ rc(nkc,n)
  common /blck12/ cw(nkc,n), cm(nkc,n)
  common /cb50/ enw(nka), ew(nkt), rn(nka), rw(nkt,nka), en(nka), e(nkt), dew(nkt), rq(nkt,nka)
  common /cb52/ ff(nkt,nka,n), fsum(n), nar(n)
  if (nmax <= nf) return
  call CW_RC_BATCH(nmax - nf, nkt, nka, .true., ff(1,1,nf+1), rq, e, kw, ka, ifeed, ddum, idum, ddum, rcd(1,nf+1), cwd(1,nf+1), ddum, ddum, idum)
  do k = nf + 1, nmax
     rc(1:2,k) = rcd(:,k)
     cw(1:2,k) = cwd(:,k)
  end do
end subroutine DRY_CW_RC_HIP

subroutine ST_COEFF_HIP_t                          ! st_coeff_t, kpp.f90:664: layers 2..nf, layer 1 keeps the default
  USE config, ONLY : lpBuxmann15alph, lpJoyce14bc
  USE global_params, ONLY : j2, j6, nf, n, nkc
  USE mistra_kpp_liq, ONLY : ST_COEFF_BATCH
  implicit none
  integer, parameter :: NSPEC = 424
  double precision :: alpha, vmean, cw, cm, sl1, sion1, theta, thetl, t, talt, p, rho, env(5, nf)
  integer :: k
  common /kpp_2tot/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  common /blck12/ cw(nkc,n), cm(nkc,n)
  common /blck17/ sl1(j2,nkc,n), sion1(j6,nkc,n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  alpha(:,1) = 0.1d0                               ! alpha(:,:) = 0.1_dp, kpp.f90:717
  do k = 2, nf
     env(:, k) = [t(k), cw(1,k), cm(1,k), sion1(13,1,k), sion1(14,1,k)]
  end do
  call ST_COEFF_BATCH(3, nf - 1, lpJoyce14bc, lpBuxmann15alph, env(1, 2), alpha(1, 2))
end subroutine ST_COEFF_HIP_t

subroutine ST_COEFF_HIP_a                          ! st_coeff_a, kpp.f90:857
  USE config, ONLY : lpJoyce14bc
  USE global_params, ONLY : j2, j6, nf, n, nkc
  USE mistra_kpp_liq, ONLY : ST_COEFF_BATCH
  implicit none
  integer, parameter :: NSPEC = 262
  double precision :: alpha, vmean, cw, cm, sl1, sion1, theta, thetl, t, talt, p, rho, env(5, nf)
  integer :: k
  common /kpp_2aer/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  common /blck12/ cw(nkc,n), cm(nkc,n)
  common /blck17/ sl1(j2,nkc,n), sion1(j6,nkc,n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  alpha(:,1) = 0.1d0                               ! alpha(:,:) = 0.1_dp, kpp.f90:905
  do k = 2, nf
     env(:, k) = [t(k), cw(1,k), cm(1,k), sion1(13,1,k), sion1(14,1,k)]
  end do
  call ST_COEFF_BATCH(2, nf - 1, lpJoyce14bc, .false., env(1, 2), alpha(1, 2))      ! (st_coeff_a has no lpBuxmann15alph branch)
end subroutine ST_COEFF_HIP_a

subroutine V_MEAN_HIP_t(tt, nmaxf)                 ! v_mean_t (tt,nmaxf), kpp.f90:1268: layers 1..nmaxf, the layers above stay 0
  USE global_params, ONLY : nf, n
  USE mistra_kpp_liq, ONLY : V_MEAN_BATCH
  implicit none
  double precision, intent(in) :: tt(n)
  integer, intent(in) :: nmaxf
  integer, parameter :: NSPEC = 424
  double precision :: alpha, vmean
  common /kpp_2tot/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  vmean(:,nmaxf+1:nf) = 0.d0                       ! vmean(:,:) = 0._dp, kpp.f90:1320
  call V_MEAN_BATCH(3, nmaxf, tt, vmean)
end subroutine V_MEAN_HIP_t

subroutine V_MEAN_HIP_a(tt, nmaxf)                 ! v_mean_a (tt,nmaxf), kpp.f90:1472
  USE global_params, ONLY : nf, n
  USE mistra_kpp_liq, ONLY : V_MEAN_BATCH
  implicit none
  double precision, intent(in) :: tt(n)
  integer, intent(in) :: nmaxf
  integer, parameter :: NSPEC = 262
  double precision :: alpha, vmean
  common /kpp_2aer/ alpha(NSPEC,nf), vmean(NSPEC,nf)
  vmean(:,nmaxf+1:nf) = 0.d0                       ! vmean(:,:) = 0._dp, kpp.f90:1525
  call V_MEAN_BATCH(2, nmaxf, tt, vmean)
end subroutine V_MEAN_HIP_a

subroutine EQUIL_CO_HIP_t(tt, nmaxf)               ! equil_co_t (tt,nmaxf), kpp.f90:2954
  USE global_params, ONLY : j6, nf, n, nkc
  USE mistra_kpp_liq, ONLY : EQUIL_CO_BATCH
  implicit none
  double precision, intent(in) :: tt(n)
  integer, intent(in) :: nmaxf
  integer, parameter :: NSPEC = 424
  double precision :: conv2, xgamma, henry, xkmt, xkef, xkeb
  common /blck13/ conv2(nkc,n)
  common /kpp_mol/ xgamma(j6,nkc,nf)
  common /kpp_ltot/ henry(NSPEC,nf), xkmt(NSPEC,nkc,nf), xkef(NSPEC,nkc,nf), xkeb(NSPEC,nkc,nf)
  call EQUIL_CO_BATCH(3, nmaxf - 1, nkc, j6, tt(2), conv2(1,2), xgamma(1,1,2), xkef(1,1,2), xkeb(1,1,2))      ! do k=2,nmaxf (kpp.f90:3020)
end subroutine EQUIL_CO_HIP_t

subroutine EQUIL_CO_HIP_a(tt, nmaxf)               ! equil_co_a (tt,nmaxf), kpp.f90:3162
  USE global_params, ONLY : j6, nf, n, nkc
  USE mistra_kpp_liq, ONLY : EQUIL_CO_BATCH
  implicit none
  double precision, intent(in) :: tt(n)
  integer, intent(in) :: nmaxf
  integer, parameter :: NSPEC = 262
  double precision :: conv2, xgamma, henry, xkmt, xkef, xkeb
  common /blck13/ conv2(nkc,n)
  common /kpp_mol/ xgamma(j6,nkc,nf)
  common /kpp_laer/ henry(NSPEC,nf), xkmt(NSPEC,nkc,nf), xkef(NSPEC,nkc,nf), xkeb(NSPEC,nkc,nf)
  call EQUIL_CO_BATCH(2, nmaxf - 1, nkc, j6, tt(2), conv2(1,2), xgamma(1,1,2), xkef(1,1,2), xkeb(1,1,2))      ! do k=2,nmaxf (kpp.f90:3228)
end subroutine EQUIL_CO_HIP_a
