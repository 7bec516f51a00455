! TEST INFRASTRUCTURE — fixture capture of whole x_drive calls of the running reference model (oracle/build_ref.sh `model`),
! for the hand-over halves of the drivers on the device (SURVEY.md §8 f2: pack, budgets, unpack).
!
! Linked with -Wl,--wrap=gas_drive_ / aer_drive_ / tot_drive_: every call kpp_driver makes (kpp.f90:4454-4467) lands here.  Written
! in Fortran because the layer's data sits in module gas_common (s1, s3 and the species index maps, allocatable) besides the
! COMMON blocks.  Around the real driver it records, for the calls selected by MISTRA_CAPTURE_DRIVE_SKIP_x / _EVERY_x / _MAX_x (per
! mechanism) or — for whole column steps in model order — by MISTRA_CAPTURE_DRIVE_SEQ_FROM / _SEQ_TO (a window of the running count of ALL
! driver calls, 148 per 10-s step):
!   before:  the driver's arguments, s1(:,k), s3(:,k), sl1(:,:,k), sion1(:,:,k), the index maps gas_m2k_x / gas_k2m_x / rad_m2k_x / rad_k2m_x
!   inside:  C = VAR | FIX as x_drive handed it to INTEGRATE_x (oracle/capture_wrap.c keeps the last one: capture_last_c_in) and the inputs of
!            its Update_RCONST_x call as MISTRA_RATES_ENV_x packs them (oracle/capture_rates_wrap.c: capture_last_env)
!   after:   RCONST and C of COMMON /GDATA_x/, s1(:,k), s3(:,k), sl1(:,:,k), sion1(:,:,k), bgs(1:2,:,k) of /budgs/ and, where k is one of
!            the budget levels il(:), bg(1:2,:,kl) of /budg/
! into MISTRA_CAPTURE_DRIVE_FILE (stream of records, see write_record).  No reference source is modified.
module capture_drive_state
  implicit none
  integer :: unit_out = 0
  logical :: inited = .false., opened = .false.
  integer :: ncall(3) = 0, nrec(3) = 0, nskip(3) = 0, nevery(3) = 1, nmax(3) = 16
  integer :: seq = 0, seq_from = 0, seq_to = 0
contains
  subroutine init()
    character(len=512) :: buf
    character(len=1), parameter :: sfx(3) = ['g', 'a', 't']
    integer :: stat, m
    inited = .true.
    call get_environment_variable('MISTRA_CAPTURE_DRIVE_FILE', buf, status=stat)
    if (stat == 0 .and. len_trim(buf) > 0) then
       open (newunit=unit_out, file=trim(buf), access='stream', form='unformatted', status='replace')
       opened = .true.
    end if
    call get_environment_variable('MISTRA_CAPTURE_DRIVE_SEQ_FROM', buf, status=stat)
    if (stat == 0) read (buf, *) seq_from
    call get_environment_variable('MISTRA_CAPTURE_DRIVE_SEQ_TO', buf, status=stat)
    if (stat == 0) read (buf, *) seq_to
    do m = 1, 3
       call get_environment_variable('MISTRA_CAPTURE_DRIVE_SKIP_'//sfx(m), buf, status=stat)
       if (stat == 0) read (buf, *) nskip(m)
       call get_environment_variable('MISTRA_CAPTURE_DRIVE_EVERY_'//sfx(m), buf, status=stat)
       if (stat == 0) read (buf, *) nevery(m)
       call get_environment_variable('MISTRA_CAPTURE_DRIVE_MAX_'//sfx(m), buf, status=stat)
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
    seq = seq + 1
    if (seq_to > 0) then
       want = opened .and. seq - 1 >= seq_from .and. seq - 1 < seq_to
       return
    end if
    want = opened .and. nrec(m) < nmax(m) .and. n >= nskip(m)
    if (want) want = mod(n - nskip(m), nevery(m)) == 0
  end function want
end module capture_drive_state

! the part shared by the three wrappers: dump one half of a record
subroutine capture_drive_layer(k, j1, j5)
  use capture_drive_state
  use gas_common, only: s1, s3
  use global_params, only: j2, j6, nkc, n
  implicit none
  integer, intent(in) :: k, j1, j5
  double precision :: sl1, sion1
  common /blck17/ sl1(j2, nkc, n), sion1(j6, nkc, n)
  write (unit_out) s1(1:j1, k), s3(1:j5, k), sl1(:, :, k), sion1(:, :, k)
end subroutine capture_drive_layer

subroutine capture_drive_budgets(k, nreact)
  use capture_drive_state
  use global_params, only: nlev, nrxn, n
  implicit none
  integer, intent(in) :: k, nreact
  double precision :: bg, bgs
  integer :: il, kl, found
  common /budg/ bg(2, nrxn, nlev), il(nlev)
  common /budgs/ bgs(2, 122, n)
  found = 0
  do kl = 1, nlev
     if (k == il(kl)) then
        found = kl
        exit
     end if
  end do
  write (unit_out) bgs(:, :, k), found
  if (found > 0) write (unit_out) bg(:, 1:nreact, found)
end subroutine capture_drive_budgets

! record: int32 {magic 'DRIV', mech, k, j1, j5, nvar, nfix, nreact, nargs}, args(nargs) [scalars in call order, then xph_rat(47)],
!         maps as int32: gas_m2k(2,j1), gas_k2m(j1), rad_m2k(2,j5), rad_k2m(j5),
!         layer before, bgs/bg before, [real driver], c_in(nspec), rconst(nreact), c_out(nspec), env(nenv), layer after, bgs/bg after
subroutine wrap_gas_drive(tkpp, dt_ch, k, yhal, yiod, yhet1, yhet2, air, h2o, xph_rat) bind(C, name="__wrap_gas_drive_")
  use capture_drive_state
  use gas_common, only: j1, j5, gas_m2k_g, gas_k2m_g, rad_m2k_g, rad_k2m_g
  implicit none
  double precision :: tkpp, dt_ch, yhal, yiod, yhet1, yhet2, air, h2o, xph_rat(47)
  integer :: k
  integer, parameter :: NVAR = 102, NFIX = 3, NREACT = 331
  double precision :: C(NVAR + NFIX), RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX, cin(NVAR + NFIX), env(74)
  common /GDATA_g/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  interface
     subroutine real_gas_drive(tkpp, dt_ch, k, yhal, yiod, yhet1, yhet2, air, h2o, xph_rat) bind(C, name="__real_gas_drive_")
       double precision :: tkpp, dt_ch, yhal, yiod, yhet1, yhet2, air, h2o, xph_rat(47)
       integer :: k
     end subroutine real_gas_drive
     subroutine capture_last_c_in(mech, out) bind(C, name="capture_last_c_in")
       integer, value :: mech
       double precision :: out(*)
     end subroutine capture_last_c_in
     subroutine capture_last_env(mech, out) bind(C, name="capture_last_env")
       integer, value :: mech
       double precision :: out(*)
     end subroutine capture_last_env
  end interface
  logical :: keep
  keep = want(1)
  if (keep) then
     write (unit_out) int(z'44524956'), 0, k, j1, j5, NVAR, NFIX, NREACT, 8 + 47
     write (unit_out) tkpp, dt_ch, yhal, yiod, yhet1, yhet2, air, h2o, xph_rat
     write (unit_out) gas_m2k_g(1:2, 1:j1), gas_k2m_g(1:j1), rad_m2k_g(1:2, 1:j5), rad_k2m_g(1:j5)
     call capture_drive_layer(k, j1, j5)
     call capture_drive_budgets(k, NREACT)
  end if
  call real_gas_drive(tkpp, dt_ch, k, yhal, yiod, yhet1, yhet2, air, h2o, xph_rat)
  if (keep) then
     call capture_last_c_in(0, cin)
     call capture_last_env(0, env)
     write (unit_out) cin, RCONST, C, env
     call capture_drive_layer(k, j1, j5)
     call capture_drive_budgets(k, NREACT)
     nrec(1) = nrec(1) + 1
  end if
end subroutine wrap_gas_drive

subroutine wrap_aer_drive(tkpp, dt_ch, k, xcvv1, xcvv2, yhal, yiod, yliq1, yliq2, yhet1, yhet2, air, h2o, xph_rat) bind(C, name="__wrap_aer_drive_")
  use capture_drive_state
  use gas_common, only: j1, j5, gas_m2k_a, gas_k2m_a, rad_m2k_a, rad_k2m_a
  implicit none
  double precision :: tkpp, dt_ch, xcvv1, xcvv2, yhal, yiod, yliq1, yliq2, yhet1, yhet2, air, h2o, xph_rat(47)
  integer :: k
  integer, parameter :: NVAR = 257, NFIX = 5, NREACT = 979
  double precision :: C(NVAR + NFIX), RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX, cin(NVAR + NFIX), env(330)
  common /GDATA_a/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  interface
     subroutine real_aer_drive(tkpp, dt_ch, k, xcvv1, xcvv2, yhal, yiod, yliq1, yliq2, yhet1, yhet2, air, h2o, xph_rat) bind(C, name="__real_aer_drive_")
       double precision :: tkpp, dt_ch, xcvv1, xcvv2, yhal, yiod, yliq1, yliq2, yhet1, yhet2, air, h2o, xph_rat(47)
       integer :: k
     end subroutine real_aer_drive
     subroutine capture_last_c_in(mech, out) bind(C, name="capture_last_c_in")
       integer, value :: mech
       double precision :: out(*)
     end subroutine capture_last_c_in
     subroutine capture_last_env(mech, out) bind(C, name="capture_last_env")
       integer, value :: mech
       double precision :: out(*)
     end subroutine capture_last_env
  end interface
  logical :: keep
  keep = want(2)
  if (keep) then
     write (unit_out) int(z'44524956'), 1, k, j1, j5, NVAR, NFIX, NREACT, 12 + 47
     write (unit_out) tkpp, dt_ch, xcvv1, xcvv2, yhal, yiod, yliq1, yliq2, yhet1, yhet2, air, h2o, xph_rat
     write (unit_out) gas_m2k_a(1:2, 1:j1), gas_k2m_a(1:j1), rad_m2k_a(1:2, 1:j5), rad_k2m_a(1:j5)
     call capture_drive_layer(k, j1, j5)
     call capture_drive_budgets(k, NREACT)
  end if
  call real_aer_drive(tkpp, dt_ch, k, xcvv1, xcvv2, yhal, yiod, yliq1, yliq2, yhet1, yhet2, air, h2o, xph_rat)
  if (keep) then
     call capture_last_c_in(1, cin)
     call capture_last_env(1, env)
     write (unit_out) cin, RCONST, C, env
     call capture_drive_layer(k, j1, j5)
     call capture_drive_budgets(k, NREACT)
     nrec(2) = nrec(2) + 1
  end if
end subroutine wrap_aer_drive

subroutine wrap_tot_drive(tkpp, dt_ch, k, xcvv1, xcvv2, xcvv3, xcvv4, yhal, yiod, yliq1, yliq2, yliq3, yliq4, yhet1, yhet2, air, h2o, xph_rat) &
     bind(C, name="__wrap_tot_drive_")
  use capture_drive_state
  use gas_common, only: j1, j5, gas_m2k_t, gas_k2m_t, rad_m2k_t, rad_k2m_t
  implicit none
  double precision :: tkpp, dt_ch, xcvv1, xcvv2, xcvv3, xcvv4, yhal, yiod, yliq1, yliq2, yliq3, yliq4, yhet1, yhet2, air, h2o, xph_rat(47)
  integer :: k
  integer, parameter :: NVAR = 417, NFIX = 7, NREACT = 1627
  double precision :: C(NVAR + NFIX), RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX, cin(NVAR + NFIX), env(544)
  common /GDATA_t/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  interface
     subroutine real_tot_drive(tkpp, dt_ch, k, xcvv1, xcvv2, xcvv3, xcvv4, yhal, yiod, yliq1, yliq2, yliq3, yliq4, yhet1, yhet2, air, h2o, xph_rat) &
          bind(C, name="__real_tot_drive_")
       double precision :: tkpp, dt_ch, xcvv1, xcvv2, xcvv3, xcvv4, yhal, yiod, yliq1, yliq2, yliq3, yliq4, yhet1, yhet2, air, h2o, xph_rat(47)
       integer :: k
     end subroutine real_tot_drive
     subroutine capture_last_c_in(mech, out) bind(C, name="capture_last_c_in")
       integer, value :: mech
       double precision :: out(*)
     end subroutine capture_last_c_in
     subroutine capture_last_env(mech, out) bind(C, name="capture_last_env")
       integer, value :: mech
       double precision :: out(*)
     end subroutine capture_last_env
  end interface
  logical :: keep
  keep = want(3)
  if (keep) then
     write (unit_out) int(z'44524956'), 2, k, j1, j5, NVAR, NFIX, NREACT, 16 + 47
     write (unit_out) tkpp, dt_ch, xcvv1, xcvv2, xcvv3, xcvv4, yhal, yiod, yliq1, yliq2, yliq3, yliq4, yhet1, yhet2, air, h2o, xph_rat
     write (unit_out) gas_m2k_t(1:2, 1:j1), gas_k2m_t(1:j1), rad_m2k_t(1:2, 1:j5), rad_k2m_t(1:j5)
     call capture_drive_layer(k, j1, j5)
     call capture_drive_budgets(k, NREACT)
  end if
  call real_tot_drive(tkpp, dt_ch, k, xcvv1, xcvv2, xcvv3, xcvv4, yhal, yiod, yliq1, yliq2, yliq3, yliq4, yhet1, yhet2, air, h2o, xph_rat)
  if (keep) then
     call capture_last_c_in(2, cin)
     call capture_last_env(2, env)
     write (unit_out) cin, RCONST, C, env
     call capture_drive_layer(k, j1, j5)
     call capture_drive_budgets(k, NREACT)
     nrec(3) = nrec(3) + 1
  end if
end subroutine wrap_tot_drive
