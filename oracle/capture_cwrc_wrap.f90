! TEST INFRASTRUCTURE — fixture capture of cw_rc / dry_cw_rc calls of the running reference model (oracle/build_ref.sh `model`), for those liq_parm
! routines on the device (SURVEY.md §8 f3).  Linked with -Wl,--wrap=cw_rc_ -Wl,--wrap=dry_cw_rc_: liq_parm's calls (kpp.f90:609, 650) land here.  For the
! calls selected by MISTRA_CAPTURE_CWRC_SKIP / _EVERY / _MAX (counted per routine) and up to MISTRA_CAPTURE_CWRC_LAYERS layers spread over the
! routine's range it records into MISTRA_CAPTURE_CWRC_FILE what the routine reads and what it leaves behind.  No reference source is modified.
! record: int32 {magic 'CWRC', routine (1 cw_rc | 2 dry_cw_rc), layers, nkt, nka, nkc, ka, ifeed, kinv, 0}, then doubles
!         rq(nkt,nka), e(nkt), kw(nka), xcryssulf, xcrysss, xdelisulf, xdeliss;  per layer: k, feu(k), cloud(1:nkc,k) (0 | 1), ff(nkt,nka,k),
!         rc(1:nkc,k), cw(1:nkc,k), cm(1:nkc,k), conv2(1:nkc,k) after the call
module capture_cwrc_state
  implicit none
  integer :: unit_out = 0, nlayers = 8
  logical :: inited = .false., opened = .false.
  integer :: ncall(2) = 0, nrec(2) = 0, nskip = 0, nevery = 1, nmax = 2
contains
  subroutine init()
    character(len=512) :: buf
    integer :: stat
    inited = .true.
    call get_environment_variable('MISTRA_CAPTURE_CWRC_FILE', buf, status=stat)
    if (stat == 0 .and. len_trim(buf) > 0) then
       open (newunit=unit_out, file=trim(buf), access='stream', form='unformatted', status='replace')
       opened = .true.
    end if
    call get_environment_variable('MISTRA_CAPTURE_CWRC_LAYERS', buf, status=stat)
    if (stat == 0) read (buf, *) nlayers
    call get_environment_variable('MISTRA_CAPTURE_CWRC_SKIP', buf, status=stat)
    if (stat == 0) read (buf, *) nskip
    call get_environment_variable('MISTRA_CAPTURE_CWRC_EVERY', buf, status=stat)
    if (stat == 0) read (buf, *) nevery
    call get_environment_variable('MISTRA_CAPTURE_CWRC_MAX', buf, status=stat)
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
  subroutine dump(routine, k0, k1)
    use config, only: ifeed
    use global_params, only: n, nka, nkt, nkc
    integer, intent(in) :: routine, k0, k1
    integer :: kw, ka, kinv, k, step, taken, klist(64), i
    double precision :: rc, cw, cm, conv2, enw, ew, rn, rw, en, e, dew, rq, ff, fsum, xm1, xm2, feu, dfddt, xm1a, xcryssulf, xcrysss, xdelisulf, xdeliss
    integer :: nar
    logical :: cloud
    common /blck06/ kw(nka), ka
    common /blck11/ rc(nkc, n)
    common /blck12/ cw(nkc, n), cm(nkc, n)
    common /blck13/ conv2(nkc, n)
    common /cb50/ enw(nka), ew(nkt), rn(nka), rw(nkt, nka), en(nka), e(nkt), dew(nkt), rq(nkt, nka)
    common /cb52/ ff(nkt, nka, n), fsum(n), nar(n)
    common /cb54/ xm1(n), xm2(n), feu(n), dfddt(n), xm1a(n)
    common /kpp_l1/ cloud(nkc, n)
    common /kpp_crys/ xcryssulf, xcrysss, xdelisulf, xdeliss
    common /kinv_i/ kinv
    taken = 0
    step = max(1, (k1 - k0 + 1) / max(1, min(nlayers, 64)))
    do k = k0, k1, step
       if (taken < min(nlayers, 64)) then
          taken = taken + 1
          klist(taken) = k
       end if
    end do
    write (unit_out) int(z'43525743'), routine, taken, nkt, nka, nkc, ka, ifeed, kinv, 0
    write (unit_out) rq, e, dble(kw), xcryssulf, xcrysss, xdelisulf, xdeliss
    do i = 1, taken
       k = klist(i)
       write (unit_out) dble(k), feu(k), merge(1.d0, 0.d0, cloud(:, k)), ff(:, :, k), rc(:, k), cw(:, k), cm(:, k), conv2(:, k)
    end do
  end subroutine dump
end module capture_cwrc_state

subroutine wrap_cw_rc(nmaxf) bind(C, name="__wrap_cw_rc_")
  use capture_cwrc_state
  implicit none
  integer :: nmaxf
  interface
     subroutine real_cw_rc(nmaxf) bind(C, name="__real_cw_rc_")
       integer :: nmaxf
     end subroutine real_cw_rc
  end interface
  logical :: keep
  keep = want(1)
  call real_cw_rc(nmaxf)      ! (cloud is an input the routine does not change)
  if (keep) call dump(1, 2, nmaxf)
end subroutine wrap_cw_rc

subroutine wrap_dry_cw_rc(nmx) bind(C, name="__wrap_dry_cw_rc_")
  use capture_cwrc_state
  use global_params, only: nf
  implicit none
  integer :: nmx
  interface
     subroutine real_dry_cw_rc(nmax) bind(C, name="__real_dry_cw_rc_")
       integer :: nmax
     end subroutine real_dry_cw_rc
  end interface
  logical :: keep
  keep = want(2)
  call real_dry_cw_rc(nmx)
  if (keep .and. nmx > nf) call dump(2, nf + 1, nmx)
end subroutine wrap_dry_cw_rc
