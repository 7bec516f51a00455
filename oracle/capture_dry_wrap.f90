! TEST INFRASTRUCTURE — fixture capture of dry_rates_g / dry_rates_a / dry_rates_t calls of the running reference model (oracle/build_ref.sh `model`),
! for those liq_parm routines on the device (SURVEY.md §8 f3).  Linked with -Wl,--wrap=dry_rates_g_ ... : liq_parm's calls (kpp.f90:651-653) land here.
! For the calls selected by MISTRA_CAPTURE_DRY_SKIP / _EVERY / _MAX (counted per routine) and up to MISTRA_CAPTURE_DRY_LAYERS layers it records into
! MISTRA_CAPTURE_DRY_FILE what the routine reads and what it leaves for the four species of its idr list.  No reference source is modified.
! record: int32 {magic 'DRYR', routine (1 g | 2 a | 3 t), k, 0}, doubles: tt(k), freep(k), rcd(1:2,k), vmean(idr,k) (g: zeros), henry(idr,k) BEFORE (a, t:
!         zeros), xkmtd(idr,1,k), xkmtd(idr,2,k), xeq(ind_HNO3,k), henry(idr,k) AFTER (a, t: zeros)
module capture_dry_state
  implicit none
  integer :: unit_out = 0, nlayers = 8
  logical :: inited = .false., opened = .false.
  integer :: ncall(3) = 0, nrec(3) = 0, nskip = 0, nevery = 1, nmax = 2
contains
  subroutine init()
    character(len=512) :: buf
    integer :: stat
    inited = .true.
    call get_environment_variable('MISTRA_CAPTURE_DRY_FILE', buf, status=stat)
    if (stat == 0 .and. len_trim(buf) > 0) then
       open (newunit=unit_out, file=trim(buf), access='stream', form='unformatted', status='replace')
       opened = .true.
    end if
    call get_environment_variable('MISTRA_CAPTURE_DRY_LAYERS', buf, status=stat)
    if (stat == 0) read (buf, *) nlayers
    call get_environment_variable('MISTRA_CAPTURE_DRY_SKIP', buf, status=stat)
    if (stat == 0) read (buf, *) nskip
    call get_environment_variable('MISTRA_CAPTURE_DRY_EVERY', buf, status=stat)
    if (stat == 0) read (buf, *) nevery
    call get_environment_variable('MISTRA_CAPTURE_DRY_MAX', buf, status=stat)
    if (stat == 0) read (buf, *) nmax
    nevery = max(1, nevery)
  end subroutine init
  logical function want(m)
    integer, intent(in) :: m
    integer :: c
    if (.not. inited) call init()
    c = ncall(m)
    ncall(m) = c + 1
    want = opened .and. nrec(m) < nmax .and. c >= nskip
    if (want) want = mod(c - nskip, nevery) == 0
    if (want) nrec(m) = nrec(m) + 1
  end function want
  subroutine pick(k1, klist, taken)
    integer, intent(in) :: k1
    integer, intent(out) :: klist(64), taken
    integer :: k, step
    taken = 0
    step = max(1, (k1 - 1) / max(1, min(nlayers, 64)))
    do k = 2, k1, step
       if (taken < min(nlayers, 64)) then
          taken = taken + 1
          klist(taken) = k
       end if
    end do
  end subroutine pick
end module capture_dry_state

subroutine wrap_dry_rates_g(tt, freep, nmx) bind(C, name="__wrap_dry_rates_g_")
  use capture_dry_state
  use global_params, only: n, nkc
  implicit none
  include 'gas_Parameters.h'
  double precision :: tt(n), freep(n)
  integer :: nmx
  double precision :: rcd, xkmtd, henry, xeq
  common /blck11/ rcd(nkc, n)
  common /kpp_dryg/ xkmtd(NSPEC, 2, n), henry(NSPEC, n), xeq(NSPEC, n)
  interface
     subroutine real_dry_rates_g(tt, freep, nmx) bind(C, name="__real_dry_rates_g_")
       double precision :: tt(*), freep(*)
       integer :: nmx
     end subroutine real_dry_rates_g
  end interface
  integer :: idr(4), klist(64), taken, i, k
  double precision :: hb(4, 64)
  logical :: keep
  idr = [ind_HNO3, ind_N2O5, ind_NH3, ind_H2SO4]
  keep = want(1)
  taken = 0
  if (keep) then
     call pick(nmx, klist, taken)
     do i = 1, taken
        hb(:, i) = henry(idr, klist(i))
     end do
  end if
  call real_dry_rates_g(tt, freep, nmx)
  do i = 1, taken
     k = klist(i)
     write (unit_out) int(z'52595244'), 1, k, 0
     write (unit_out) tt(k), freep(k), rcd(1:2, k), 0.d0, 0.d0, 0.d0, 0.d0, hb(:, i), xkmtd(idr, 1, k), xkmtd(idr, 2, k), xeq(ind_HNO3, k), henry(idr, k)
  end do
end subroutine wrap_dry_rates_g

subroutine wrap_dry_rates_a(freep, nmaxf) bind(C, name="__wrap_dry_rates_a_")
  use capture_dry_state
  use global_params, only: nf, n, nkc
  implicit none
  include 'aer_Parameters.h'
  double precision :: freep(n)
  integer :: nmaxf
  double precision :: rcd, xkmtd, xeq, alpha, vmean, theta, thetl, t, talt, p, rho
  common /blck11/ rcd(nkc, n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_2aer/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  common /kpp_drya/ xkmtd(NSPEC, 2, nf), xeq(NSPEC, nf)
  interface
     subroutine real_dry_rates_a(freep, nmaxf) bind(C, name="__real_dry_rates_a_")
       double precision :: freep(*)
       integer :: nmaxf
     end subroutine real_dry_rates_a
  end interface
  integer :: idr(4), klist(64), taken, i, k
  logical :: keep
  idr = [ind_HNO3, ind_N2O5, ind_NH3, ind_H2SO4]
  keep = want(2)
  call real_dry_rates_a(freep, nmaxf)
  if (keep) then
     call pick(nmaxf, klist, taken)
     do i = 1, taken
        k = klist(i)
        write (unit_out) int(z'52595244'), 2, k, 0
        write (unit_out) t(k), freep(k), rcd(1:2, k), vmean(idr, k), 0.d0, 0.d0, 0.d0, 0.d0, xkmtd(idr, 1, k), xkmtd(idr, 2, k), xeq(ind_HNO3, k), 0.d0, 0.d0, 0.d0, 0.d0
     end do
  end if
end subroutine wrap_dry_rates_a

subroutine wrap_dry_rates_t(freep, nmaxf) bind(C, name="__wrap_dry_rates_t_")
  use capture_dry_state
  use global_params, only: nf, n, nkc
  implicit none
  include 'tot_Parameters.h'
  double precision :: freep(n)
  integer :: nmaxf
  double precision :: rcd, xkmtd, xeq, alpha, vmean, theta, thetl, t, talt, p, rho
  common /blck11/ rcd(nkc, n)
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_2tot/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  common /kpp_dryt/ xkmtd(NSPEC, 2, nf), xeq(NSPEC, nf)
  interface
     subroutine real_dry_rates_t(freep, nmaxf) bind(C, name="__real_dry_rates_t_")
       double precision :: freep(*)
       integer :: nmaxf
     end subroutine real_dry_rates_t
  end interface
  integer :: idr(4), klist(64), taken, i, k
  logical :: keep
  idr = [ind_HNO3, ind_N2O5, ind_NH3, ind_H2SO4]
  keep = want(3)
  call real_dry_rates_t(freep, nmaxf)
  if (keep) then
     call pick(nmaxf, klist, taken)
     do i = 1, taken
        k = klist(i)
        write (unit_out) int(z'52595244'), 3, k, 0
        write (unit_out) t(k), freep(k), rcd(1:2, k), vmean(idr, k), 0.d0, 0.d0, 0.d0, 0.d0, xkmtd(idr, 1, k), xkmtd(idr, 2, k), xeq(ind_HNO3, k), 0.d0, 0.d0, 0.d0, 0.d0
     end do
  end if
end subroutine wrap_dry_rates_t
