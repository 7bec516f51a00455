! TEST INFRASTRUCTURE — fixture capture of fast_k_mt_a / fast_k_mt_t calls of the running reference model (oracle/build_ref.sh
! `model`), for the mass-transfer coefficients on the device (SURVEY.md §8 f3, first slice).
!
! Linked with -Wl,--wrap=fast_k_mt_a_ / fast_k_mt_t_: liq_parm's calls (kpp.f90:617,637) land here.  For the calls selected by
! MISTRA_CAPTURE_KMT_SKIP_x / _EVERY_x / _MAX_x (x = a | t) and, inside them, up to MISTRA_CAPTURE_KMT_LAYERS layers with an active
! bin (cm > 0), it records what the routine READS for that layer — the particle spectrum ff(:,:,k), cw(:,k), cm(:,k), freep(k),
! alpha(:,k), vmean(:,k), t(k), p(k) (the terminal velocity's arguments), plus once per record rq, kw, ka, ifeed, nkc_l — and xkmt(:,:,k)
! and the LWC-weighted sedimentation velocity vt(:,k) of /kpp_vt/ before and after the real call, into MISTRA_CAPTURE_KMT_FILE.  No
! reference source is modified.
! record: int32 {magic 'KMTC', variant (1 a | 2 t), k, nspec, nka, nkt, nkc, ka, ifeed, nkc_l}, int32 kw(nka),
!         doubles rq(nkt,nka), ff(nkt,nka), cw(nkc), cm(nkc), freep, alpha(nspec), vmean(nspec), xkmt_before(nspec,nkc), xkmt_after(nspec,nkc),
!         t(k), p(k), vt_before(nkc), vt_after(nkc)
module capture_kmt_state
  implicit none
  integer :: unit_out = 0, nlayers = 4
  logical :: inited = .false., opened = .false.
  integer :: ncall(2) = 0, nrec(2) = 0, nskip(2) = 0, nevery(2) = 1, nmax(2) = 4
contains
  subroutine init()
    character(len=512) :: buf
    character(len=1), parameter :: sfx(2) = ['a', 't']
    integer :: stat, m
    inited = .true.
    call get_environment_variable('MISTRA_CAPTURE_KMT_FILE', buf, status=stat)
    if (stat == 0 .and. len_trim(buf) > 0) then
       open (newunit=unit_out, file=trim(buf), access='stream', form='unformatted', status='replace')
       opened = .true.
    end if
    call get_environment_variable('MISTRA_CAPTURE_KMT_LAYERS', buf, status=stat)
    if (stat == 0) read (buf, *) nlayers
    do m = 1, 2
       call get_environment_variable('MISTRA_CAPTURE_KMT_SKIP_'//sfx(m), buf, status=stat)
       if (stat == 0) read (buf, *) nskip(m)
       call get_environment_variable('MISTRA_CAPTURE_KMT_EVERY_'//sfx(m), buf, status=stat)
       if (stat == 0) read (buf, *) nevery(m)
       call get_environment_variable('MISTRA_CAPTURE_KMT_MAX_'//sfx(m), buf, status=stat)
       if (stat == 0) read (buf, *) nmax(m)
       nevery(m) = max(1, nevery(m))
    end do
  end subroutine init
  logical function want(m)
    integer, intent(in) :: m
    integer :: n
    if (.not. inited) call init()
    n = ncall(m)
    ncall(m) = n + 1
    want = opened .and. nrec(m) < nmax(m) .and. n >= nskip(m)
    if (want) want = mod(n - nskip(m), nevery(m)) == 0
  end function want
end module capture_kmt_state

subroutine wrap_fast_k_mt_a(freep, box, n_bl) bind(C, name="__wrap_fast_k_mt_a_")
  use capture_kmt_state
  use config, only: ifeed, nkc_l
  use global_params, only: nf, n, nka, nkt, nkc
  implicit none
  double precision :: freep(n)
  logical :: box
  integer :: n_bl
  integer, parameter :: NSPEC = 262
  integer :: kw, ka, nar, k, taken, klist(64), i
  double precision :: cw, cm, enw, ew, rn, rw, en, e, dew, rq, ff, fsum, alpha, vmean, henry, xkmt, xkef, xkeb
  common /blck06/ kw(nka), ka
  common /blck12/ cw(nkc, n), cm(nkc, n)
  common /cb50/ enw(nka), ew(nkt), rn(nka), rw(nkt, nka), en(nka), e(nkt), dew(nkt), rq(nkt, nka)
  common /cb52/ ff(nkt, nka, n), fsum(n), nar(n)
  common /kpp_2aer/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  common /kpp_laer/ henry(NSPEC, nf), xkmt(NSPEC, nkc, nf), xkef(NSPEC, nkc, nf), xkeb(NSPEC, nkc, nf)
  double precision :: theta, thetl, t, talt, p, rho, vt, vd, vdm
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_vt/ vt(nkc, nf), vd(nkt, nka), vdm(nkc)
  double precision, allocatable :: before(:, :, :), vt_before(:, :)
  interface
     subroutine real_fast_k_mt_a(freep, box, n_bl) bind(C, name="__real_fast_k_mt_a_")
       double precision :: freep(*)
       logical :: box
       integer :: n_bl
     end subroutine real_fast_k_mt_a
  end interface
  logical :: keep
  keep = want(1)
  taken = 0
  if (keep) then
     do k = 2, nf      ! layers with an active bin, spread over the column
        if (any(cm(1:nkc_l, k) > 0.d0) .and. taken < min(nlayers, 64)) then
           if (taken == 0 .or. mod(k, 13) == 0) then
              taken = taken + 1
              klist(taken) = k
           end if
        end if
     end do
     allocate (before(NSPEC, nkc, taken), vt_before(nkc, taken))
     do i = 1, taken
        before(:, :, i) = xkmt(:, :, klist(i))
        vt_before(:, i) = vt(:, klist(i))
     end do
  end if
  call real_fast_k_mt_a(freep, box, n_bl)
  if (keep) then
     do i = 1, taken
        k = klist(i)
        write (unit_out) int(z'4B4D5443'), 1, k, NSPEC, nka, nkt, nkc, ka, ifeed, nkc_l
        write (unit_out) kw
        write (unit_out) rq, ff(:, :, k), cw(:, k), cm(:, k), freep(k), alpha(:, k), vmean(:, k), before(:, :, i), xkmt(:, :, k), &
             t(k), p(k), vt_before(:, i), vt(:, k)
     end do
     if (taken > 0) nrec(1) = nrec(1) + 1
  end if
end subroutine wrap_fast_k_mt_a

subroutine wrap_fast_k_mt_t(freep, box, n_bl) bind(C, name="__wrap_fast_k_mt_t_")
  use capture_kmt_state
  use config, only: ifeed, nkc_l
  use global_params, only: nf, n, nka, nkt, nkc
  implicit none
  double precision :: freep(n)
  logical :: box
  integer :: n_bl
  integer, parameter :: NSPEC = 424
  integer :: kw, ka, nar, k, taken, klist(64), i
  double precision :: cw, cm, enw, ew, rn, rw, en, e, dew, rq, ff, fsum, alpha, vmean, henry, xkmt, xkef, xkeb
  common /blck06/ kw(nka), ka
  common /blck12/ cw(nkc, n), cm(nkc, n)
  common /cb50/ enw(nka), ew(nkt), rn(nka), rw(nkt, nka), en(nka), e(nkt), dew(nkt), rq(nkt, nka)
  common /cb52/ ff(nkt, nka, n), fsum(n), nar(n)
  common /kpp_2tot/ alpha(NSPEC, nf), vmean(NSPEC, nf)
  common /kpp_ltot/ henry(NSPEC, nf), xkmt(NSPEC, nkc, nf), xkef(NSPEC, nkc, nf), xkeb(NSPEC, nkc, nf)
  double precision :: theta, thetl, t, talt, p, rho, vt, vd, vdm
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  common /kpp_vt/ vt(nkc, nf), vd(nkt, nka), vdm(nkc)
  double precision, allocatable :: before(:, :, :), vt_before(:, :)
  interface
     subroutine real_fast_k_mt_t(freep, box, n_bl) bind(C, name="__real_fast_k_mt_t_")
       double precision :: freep(*)
       logical :: box
       integer :: n_bl
     end subroutine real_fast_k_mt_t
  end interface
  logical :: keep
  keep = want(2)
  taken = 0
  if (keep) then
     do k = 2, nf
        if (any(cm(1:nkc, k) > 0.d0) .and. taken < min(nlayers, 64)) then
           if (taken == 0 .or. mod(k, 13) == 0) then
              taken = taken + 1
              klist(taken) = k
           end if
        end if
     end do
     allocate (before(NSPEC, nkc, taken), vt_before(nkc, taken))
     do i = 1, taken
        before(:, :, i) = xkmt(:, :, klist(i))
        vt_before(:, i) = vt(:, klist(i))
     end do
  end if
  call real_fast_k_mt_t(freep, box, n_bl)
  if (keep) then
     do i = 1, taken
        k = klist(i)
        write (unit_out) int(z'4B4D5443'), 2, k, NSPEC, nka, nkt, nkc, ka, ifeed, nkc
        write (unit_out) kw
        write (unit_out) rq, ff(:, :, k), cw(:, k), cm(:, k), freep(k), alpha(:, k), vmean(:, k), before(:, :, i), xkmt(:, :, k), &
             t(k), p(k), vt_before(:, i), vt(:, k)
     end do
     if (taken > 0) nrec(2) = nrec(2) + 1
  end if
end subroutine wrap_fast_k_mt_t
