! TEST INFRASTRUCTURE — fixture capture of henry_a / henry_t / equil_co_a / equil_co_t / v_mean_a / v_mean_t / st_coeff_a / st_coeff_t calls of the running
! reference model
! (oracle/build_ref.sh `model`), for those liq_parm kernels on the device (SURVEY.md §8 f3).
!
! Linked with -Wl,--wrap=henry_a_ ... equil_co_t_, v_mean_a_, v_mean_t_: liq_parm's calls (kpp.f90:612-616, 632-636) land here.  For the calls selected by
! MISTRA_CAPTURE_LIQ_SKIP / _EVERY / _MAX (counted per routine) and, inside them, up to MISTRA_CAPTURE_LIQ_LAYERS layers spread over
! the column, it records what the routine READS for that layer and what it leaves behind, into MISTRA_CAPTURE_LIQ_FILE:
!   henry_x     tt(k)                                      -> henry(:,k)
!   v_mean_x    tt(k)                                      -> vmean(:,k)
!   st_coeff_x  t(k), cw(1,k), cm(1,k), sion1(13:14,1,k), the switches lpJoyce14bc, lpBuxmann15alph    -> alpha(:,k)
!   equil_co_x  tt(k), conv2(:,k), xgamma(:,:,k)           -> xkef(:,:,k), xkeb(:,:,k), both also BEFORE the call (entries of species the
!                                                             routine does not set keep what they held)
! No reference source is modified.
! record: int32 {magic 'LIQC', routine (1 henry_a | 2 henry_t | 3 equil_co_a | 4 equil_co_t | 5 v_mean_a | 6 v_mean_t | 7 st_coeff_a | 8 st_coeff_t), k, nspec, nkc, j6},
!         then doubles
!         routines 1, 2:  tt, henry(nspec)            routines 5, 6:  tt, vmean(nspec)
!         routines 7, 8:  t, cw1, cm1, sion1_13, sion1_14, lpJoyce14bc (0 | 1), lpBuxmann15alph (0 | 1), alpha(nspec)
!         routines 3, 4:  tt, conv2(nkc), xgamma(j6,nkc), xkef_before(nspec,nkc), xkeb_before(nspec,nkc), xkef(nspec,nkc), xkeb(nspec,nkc)
module capture_liq_state
  implicit none
  integer :: unit_out = 0, nlayers = 6
  logical :: inited = .false., opened = .false.
  integer :: ncall(8) = 0, nrec(8) = 0, nskip = 0, nevery = 1, nmax = 2
contains
  subroutine init()
    character(len=512) :: buf
    integer :: stat
    inited = .true.
    call get_environment_variable('MISTRA_CAPTURE_LIQ_FILE', buf, status=stat)
    if (stat == 0 .and. len_trim(buf) > 0) then
       open (newunit=unit_out, file=trim(buf), access='stream', form='unformatted', status='replace')
       opened = .true.
    end if
    call get_environment_variable('MISTRA_CAPTURE_LIQ_LAYERS', buf, status=stat)
    if (stat == 0) read (buf, *) nlayers
    call get_environment_variable('MISTRA_CAPTURE_LIQ_SKIP', buf, status=stat)
    if (stat == 0) read (buf, *) nskip
    call get_environment_variable('MISTRA_CAPTURE_LIQ_EVERY', buf, status=stat)
    if (stat == 0) read (buf, *) nevery
    call get_environment_variable('MISTRA_CAPTURE_LIQ_MAX', buf, status=stat)
    if (stat == 0) read (buf, *) nmax
    nevery = max(1, nevery)
  end subroutine init
  logical function want(m)
    integer, intent(in) :: m
    integer :: n
    if (.not. inited) call init()
    n = ncall(m)
    ncall(m) = n + 1
    want = opened .and. nrec(m) < nmax .and. n >= nskip
    if (want) want = mod(n - nskip, nevery) == 0
    if (want) nrec(m) = nrec(m) + 1
  end function want
  ! layers 2, 2 + step, ... up to nmaxf, at most nlayers of them
  subroutine pick(nmaxf, klist, taken)
    integer, intent(in) :: nmaxf
    integer, intent(out) :: klist(64), taken
    integer :: k, step
    taken = 0
    step = max(1, (nmaxf - 1) / max(1, min(nlayers, 64)))
    do k = 2, nmaxf, step
       if (taken < min(nlayers, 64)) then
          taken = taken + 1
          klist(taken) = k
       end if
    end do
  end subroutine pick
end module capture_liq_state

subroutine wrap_henry_a(tt, nmaxf) bind(C, name="__wrap_henry_a_")
  use capture_liq_state
  use global_params, only: nf, n, nkc, j6
  implicit none
  double precision :: tt(n)
  integer :: nmaxf
  integer, parameter :: NSPEC = 262
  double precision :: henry, xkmt, xkef, xkeb
  common /kpp_laer/ henry(NSPEC, nf), xkmt(NSPEC, nkc, nf), xkef(NSPEC, nkc, nf), xkeb(NSPEC, nkc, nf)
  interface
     subroutine real_henry_a(tt, nmaxf) bind(C, name="__real_henry_a_")
       double precision :: tt(*)
       integer :: nmaxf
     end subroutine real_henry_a
  end interface
  integer :: klist(64), taken, i
  logical :: keep
  keep = want(1)
  call real_henry_a(tt, nmaxf)
  if (keep) then
     call pick(nmaxf, klist, taken)
     do i = 1, taken
        write (unit_out) int(z'4C495143'), 1, klist(i), NSPEC, nkc, j6
        write (unit_out) tt(klist(i)), henry(:, klist(i))
     end do
  end if
end subroutine wrap_henry_a

subroutine wrap_henry_t(tt, nmaxf) bind(C, name="__wrap_henry_t_")
  use capture_liq_state
  use global_params, only: nf, n, nkc, j6
  implicit none
  double precision :: tt(n)
  integer :: nmaxf
  integer, parameter :: NSPEC = 424
  double precision :: henry, xkmt, xkef, xkeb
  common /kpp_ltot/ henry(NSPEC, nf), xkmt(NSPEC, nkc, nf), xkef(NSPEC, nkc, nf), xkeb(NSPEC, nkc, nf)
  interface
     subroutine real_henry_t(tt, nmaxf) bind(C, name="__real_henry_t_")
       double precision :: tt(*)
       integer :: nmaxf
     end subroutine real_henry_t
  end interface
  integer :: klist(64), taken, i
  logical :: keep
  keep = want(2)
  call real_henry_t(tt, nmaxf)
  if (keep) then
     call pick(nmaxf, klist, taken)
     do i = 1, taken
        write (unit_out) int(z'4C495143'), 2, klist(i), NSPEC, nkc, j6
        write (unit_out) tt(klist(i)), henry(:, klist(i))
     end do
  end if
end subroutine wrap_henry_t

subroutine wrap_equil_co_a(tt, nmaxf) bind(C, name="__wrap_equil_co_a_")
  use capture_liq_state
  use global_params, only: nf, n, nkc, j6
  implicit none
  double precision :: tt(n)
  integer :: nmaxf
  integer, parameter :: NSPEC = 262
  double precision :: henry, xkmt, xkef, xkeb, conv2, xgamma
  common /kpp_laer/ henry(NSPEC, nf), xkmt(NSPEC, nkc, nf), xkef(NSPEC, nkc, nf), xkeb(NSPEC, nkc, nf)
  common /blck13/ conv2(nkc, n)
  common /kpp_mol/ xgamma(j6, nkc, nf)
  interface
     subroutine real_equil_co_a(tt, nmaxf) bind(C, name="__real_equil_co_a_")
       double precision :: tt(*)
       integer :: nmaxf
     end subroutine real_equil_co_a
  end interface
  integer :: klist(64), taken, i, k
  logical :: keep
  double precision, allocatable :: bf(:, :, :), bb(:, :, :)
  keep = want(3)
  taken = 0
  if (keep) then
     call pick(nmaxf, klist, taken)
     allocate (bf(NSPEC, nkc, taken), bb(NSPEC, nkc, taken))
     do i = 1, taken
        bf(:, :, i) = xkef(:, :, klist(i))
        bb(:, :, i) = xkeb(:, :, klist(i))
     end do
  end if
  call real_equil_co_a(tt, nmaxf)
  do i = 1, taken
     k = klist(i)
     write (unit_out) int(z'4C495143'), 3, k, NSPEC, nkc, j6
     write (unit_out) tt(k), conv2(:, k), xgamma(:, :, k), bf(:, :, i), bb(:, :, i), xkef(:, :, k), xkeb(:, :, k)
  end do
end subroutine wrap_equil_co_a

subroutine wrap_equil_co_t(tt, nmaxf) bind(C, name="__wrap_equil_co_t_")
  use capture_liq_state
  use global_params, only: nf, n, nkc, j6
  implicit none
  double precision :: tt(n)
  integer :: nmaxf
  integer, parameter :: NSPEC = 424
  double precision :: henry, xkmt, xkef, xkeb, conv2, xgamma
  common /kpp_ltot/ henry(NSPEC, nf), xkmt(NSPEC, nkc, nf), xkef(NSPEC, nkc, nf), xkeb(NSPEC, nkc, nf)
  common /blck13/ conv2(nkc, n)
  common /kpp_mol/ xgamma(j6, nkc, nf)
  interface
     subroutine real_equil_co_t(tt, nmaxf) bind(C, name="__real_equil_co_t_")
       double precision :: tt(*)
       integer :: nmaxf
     end subroutine real_equil_co_t
  end interface
  integer :: klist(64), taken, i, k
  logical :: keep
  double precision, allocatable :: bf(:, :, :), bb(:, :, :)
  keep = want(4)
  taken = 0
  if (keep) then
     call pick(nmaxf, klist, taken)
     allocate (bf(NSPEC, nkc, taken), bb(NSPEC, nkc, taken))
     do i = 1, taken
        bf(:, :, i) = xkef(:, :, klist(i))
        bb(:, :, i) = xkeb(:, :, klist(i))
     end do
  end if
  call real_equil_co_t(tt, nmaxf)
  do i = 1, taken
     k = klist(i)
     write (unit_out) int(z'4C495143'), 4, k, NSPEC, nkc, j6
     write (unit_out) tt(k), conv2(:, k), xgamma(:, :, k), bf(:, :, i), bb(:, :, i), xkef(:, :, k), xkeb(:, :, k)
  end do
end subroutine wrap_equil_co_t

subroutine wrap_v_mean_a(tt, nmaxf) bind(C, name="__wrap_v_mean_a_")
  use capture_liq_state
  use global_params, only: nf, n, nkc, j6
  implicit none
  double precision :: tt(n)
  integer :: nmaxf
  integer, parameter :: NSPEC = 262
  double precision :: alpha, vmean
  common /kpp_2aer/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  interface
     subroutine real_v_mean_a(tt, nmaxf) bind(C, name="__real_v_mean_a_")
       double precision :: tt(*)
       integer :: nmaxf
     end subroutine real_v_mean_a
  end interface
  integer :: klist(64), taken, i
  logical :: keep
  keep = want(5)
  call real_v_mean_a(tt, nmaxf)
  if (keep) then
     call pick(nmaxf, klist, taken)
     do i = 1, taken
        write (unit_out) int(z'4C495143'), 5, klist(i), NSPEC, nkc, j6
        write (unit_out) tt(klist(i)), vmean(:, klist(i))
     end do
  end if
end subroutine wrap_v_mean_a

subroutine wrap_v_mean_t(tt, nmaxf) bind(C, name="__wrap_v_mean_t_")
  use capture_liq_state
  use global_params, only: nf, n, nkc, j6
  implicit none
  double precision :: tt(n)
  integer :: nmaxf
  integer, parameter :: NSPEC = 424
  double precision :: alpha, vmean
  common /kpp_2tot/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  interface
     subroutine real_v_mean_t(tt, nmaxf) bind(C, name="__real_v_mean_t_")
       double precision :: tt(*)
       integer :: nmaxf
     end subroutine real_v_mean_t
  end interface
  integer :: klist(64), taken, i
  logical :: keep
  keep = want(6)
  call real_v_mean_t(tt, nmaxf)
  if (keep) then
     call pick(nmaxf, klist, taken)
     do i = 1, taken
        write (unit_out) int(z'4C495143'), 6, klist(i), NSPEC, nkc, j6
        write (unit_out) tt(klist(i)), vmean(:, klist(i))
     end do
  end if
end subroutine wrap_v_mean_t

subroutine wrap_st_coeff_a() bind(C, name="__wrap_st_coeff_a_")
  use capture_liq_state
  use config, only: lpJoyce14bc, lpBuxmann15alph
  use global_params, only: nf, n, nkc, j2, j6
  implicit none
  integer, parameter :: NSPEC = 262
  double precision :: alpha, vmean, cw, cm, sl1, sion1, theta, thetl, t, talt, p, rho
  common /kpp_2aer/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  common /blck12/ cw(nkc, n), cm(nkc, n)
  common /blck17/ sl1(j2, nkc, n), sion1(j6, nkc, n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  interface
     subroutine real_st_coeff_a() bind(C, name="__real_st_coeff_a_")
     end subroutine real_st_coeff_a
  end interface
  integer :: klist(64), taken, i, k
  logical :: keep
  keep = want(7)
  call real_st_coeff_a()
  if (keep) then
     call pick(nf, klist, taken)
     do i = 1, taken
        k = klist(i)
        write (unit_out) int(z'4C495143'), 7, k, NSPEC, nkc, j6
        write (unit_out) t(k), cw(1, k), cm(1, k), sion1(13, 1, k), sion1(14, 1, k), merge(1.d0, 0.d0, lpJoyce14bc), merge(1.d0, 0.d0, lpBuxmann15alph), alpha(:, k)
     end do
  end if
end subroutine wrap_st_coeff_a

subroutine wrap_st_coeff_t() bind(C, name="__wrap_st_coeff_t_")
  use capture_liq_state
  use config, only: lpJoyce14bc, lpBuxmann15alph
  use global_params, only: nf, n, nkc, j2, j6
  implicit none
  integer, parameter :: NSPEC = 424
  double precision :: alpha, vmean, cw, cm, sl1, sion1, theta, thetl, t, talt, p, rho
  common /kpp_2tot/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  common /blck12/ cw(nkc, n), cm(nkc, n)
  common /blck17/ sl1(j2, nkc, n), sion1(j6, nkc, n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  interface
     subroutine real_st_coeff_t() bind(C, name="__real_st_coeff_t_")
     end subroutine real_st_coeff_t
  end interface
  integer :: klist(64), taken, i, k
  logical :: keep
  keep = want(8)
  call real_st_coeff_t()
  if (keep) then
     call pick(nf, klist, taken)
     do i = 1, taken
        k = klist(i)
        write (unit_out) int(z'4C495143'), 8, k, NSPEC, nkc, j6
        write (unit_out) t(k), cw(1, k), cm(1, k), sion1(13, 1, k), sion1(14, 1, k), merge(1.d0, 0.d0, lpJoyce14bc), merge(1.d0, 0.d0, lpBuxmann15alph), alpha(:, k)
     end do
  end if
end subroutine wrap_st_coeff_t
