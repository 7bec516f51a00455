! Fortran side of the drop-in boundary: the reference's own subroutine surface,
!     SUBROUTINE INTEGRATE_g / INTEGRATE_a / INTEGRATE_t (TIN, TOUT)          gas.f:710 | aer.f:1408 | tot.f:2812
! implemented on top of the C ABI (include/mistra_chem.h) through ISO_C_BINDING, plus the batched form a two-pass
! kpp_driver calls once per mechanism and 10-s step (INTEGRATION.md):
!     SUBROUTINE INTEGRATE_BATCH_g / _a / _t (NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
!
! Same names, same two REAL*8 arguments by reference, same data path: everything else travels through
! COMMON /GDATA_x/, whose member order is restated below from gas_Global.h:29-58 (aer_Global.h, tot_Global.h alike):
!     C(NSPEC) [= VAR(NVAR) followed by FIX(NFIX)], RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
! A maintainer drops this file into src/ and removes (or renames) the three generated INTEGRATE_x routines; x_drive,
! Update_RCONST_x, the budgets and kpp_driver stay untouched (INTEGRATION.md shows the link line and the
! link-time alternative that needs no source edit).  Behaviour kept from the reference: VAR is advanced in place,
! TIN returns the exit time, STEPMIN the last step, RTOL/ATOL are (re)set to 1e-3 / 1e-25, an unsuccessful
! integration writes the lines of ros_ErrorMsg_x (gas.f:1474-1509) and of INTEGRATE_x (gas.f:764-767) to unit 6 and the
! model carries on; so does ros_PrepareMatrix_x's 'Warning: LU Decomposition returned ising =' line with the row of the zero
! pivot (gas.f:1456; the kernel records the rows, mistra_chem_singular_rows hands them out), once per failed decomposition.
module mistra_chem_c_api
  use iso_c_binding
  implicit none
  interface
     function mistra_chem_integrate_common_status(mech, gdata, tin, tout, ierr, t_err, h_err, nsng) &
          bind(C, name="mistra_chem_integrate_common_status") result(rc)
       import :: c_int, c_ptr, c_double, c_int32_t
       integer(c_int), value :: mech
       type(c_ptr), value :: gdata
       real(c_double) :: tin, tout, t_err, h_err
       integer(c_int32_t) :: ierr, nsng
       integer(c_int) :: rc
     end function mistra_chem_integrate_common_status
     function mistra_chem_integrate_ex(mech, ncell, var_in, fix, rconst, tin, tout, var_out, ierr, stats, t_h) &
          bind(C, name="mistra_chem_integrate_ex") result(rc)
       import :: c_int, c_double, c_int32_t
       integer(c_int), value :: mech, ncell
       real(c_double), value :: tin, tout
       real(c_double) :: var_in(*), fix(*), rconst(*), var_out(*), t_h(*)
       integer(c_int32_t) :: ierr(*), stats(*)
       integer(c_int) :: rc
     end function mistra_chem_integrate_ex
     function mistra_chem_integrate_env_ex(mech, ncell, var_in, fix, env, tin, tout, var_out, ierr, stats, t_h) &
          bind(C, name="mistra_chem_integrate_env_ex") result(rc)
       import :: c_int, c_double, c_int32_t
       integer(c_int), value :: mech, ncell
       real(c_double), value :: tin, tout
       real(c_double) :: var_in(*), fix(*), env(*), var_out(*), t_h(*)
       integer(c_int32_t) :: ierr(*), stats(*)
       integer(c_int) :: rc
     end function mistra_chem_integrate_env_ex
     function mistra_chem_init_devices(n_devices, device_ids) bind(C, name="mistra_chem_init_devices") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: n_devices
       type(c_ptr), value :: device_ids
       integer(c_int) :: rc
     end function mistra_chem_init_devices
     function mistra_chem_singular_rows(mech, cell, rows8) bind(C, name="mistra_chem_singular_rows") result(rc)
       import :: c_int, c_int32_t
       integer(c_int), value :: mech, cell
       integer(c_int32_t) :: rows8(8)
       integer(c_int) :: rc
     end function mistra_chem_singular_rows
     function mistra_chem_last_error() bind(C, name="mistra_chem_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function mistra_chem_last_error
  end interface
contains
  subroutine mistra_chem_fail(where)
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: txt(:)
    integer :: i
    call c_f_pointer(mistra_chem_last_error(), txt, [256])
    write (0, '(3a)', advance='no') ' mistra_chem: ', where, ' failed: '
    do i = 1, 256
       if (txt(i) == c_null_char) exit
       write (0, '(a)', advance='no') txt(i)
    end do
    write (0, *)
    error stop 'mistra_chem: GPU integrator unavailable (there is no CPU fallback)'
  end subroutine mistra_chem_fail

  ! Use the first n GPUs of the node for the batched calls (one block of cells and one host thread per device inside the
  ! library).  Optional: without it the first call initialises device MISTRA_CHEM_DEVICE (default 0).
  subroutine mistra_chem_use_devices(n)
    integer, intent(in) :: n
    if (mistra_chem_init_devices(int(n, c_int), c_null_ptr) /= 0) call mistra_chem_fail('mistra_chem_init_devices')
  end subroutine mistra_chem_use_devices

  ! The reference's messages for an unsuccessful integration, statement by statement: ros_ErrorMsg_x (gas.f:1474-1509)
  ! and the PRINT of INTEGRATE_x (gas.f:764-767).  sfx = 'g' | 'a' | 't'.
  ! mech, cell (0-based index into the last batch): where the rows of the zero pivots are asked for
  subroutine mistra_chem_report(sfx, code, t, h, tin, nsng, mech, cell)
    character(len=1), intent(in) :: sfx
    integer, intent(in) :: code, nsng, mech, cell
    double precision, intent(in) :: t, h, tin
    integer :: i
    integer(c_int32_t) :: rows(8)
    if (nsng > 0) then
       rows = 0
       if (mistra_chem_singular_rows(int(mech, c_int), int(cell, c_int), rows) /= 0) call mistra_chem_fail('mistra_chem_singular_rows')
       do i = 1, nsng       ! (the kernel keeps the first eight rows of a call; a ninth failed decomposition repeats the eighth)
          print *, 'Warning: LU Decomposition returned ising = ', int(rows(min(i, 8)))
       end do
    end if
    if (code >= 0) return
    write (6, *) 'Forced exit from Rosenbrock_'//sfx//' due to the following error:'
    if (code == -1) then
       write (6, *) '--> Improper value for maximal no of steps'
    else if (code == -2) then
       write (6, *) '--> Selected Rosenbrock method not implemented'
    else if (code == -3) then
       write (6, *) '--> Hmin/Hmax/Hstart must be positive'
    else if (code == -4) then
       write (6, *) '--> FacMin/FacMax/FacRej must be positive'
    else if (code == -5) then
       write (6, *) '--> Improper tolerance values'
    else if (code == -6) then
       write (6, *) '--> No of steps exceeds maximum bound'
    else if (code == -7) then
       write (6, *) '--> Step size too small: T + 10*H = T', ' or H < Roundoff'
    else if (code == -8) then
       write (6, *) '--> Matrix is repeatedly singular'
    else
       write (6, 102) 'Unknown Error code: ', code
    end if
102 format('       ', A, I4)
    write (6, 103) t, h
103 format('        T=', E15.7, ' and H=', E15.7)
    print *, 'Rosenbrock: Unsucessful step at T=', tin, ' (IERR=', code, ')'
  end subroutine mistra_chem_report

  ! the three one-cell routines and the three batched ones differ in sizes only
  subroutine integrate_one(mech, sfx, gdata, TIN, TOUT)
    integer, intent(in) :: mech
    character(len=1), intent(in) :: sfx
    type(c_ptr), intent(in) :: gdata
    real(c_double) :: TIN, TOUT
    integer(c_int32_t) :: ierr, nsng
    real(c_double) :: t_err, h_err, tin_in
    tin_in = TIN
    if (mistra_chem_integrate_common_status(int(mech, c_int), gdata, TIN, TOUT, ierr, t_err, h_err, nsng) /= 0) &
         call mistra_chem_fail('INTEGRATE_'//sfx)
    if (ierr < 0 .or. nsng > 0) call mistra_chem_report(sfx, int(ierr), t_err, h_err, tin_in, int(nsng), mech, 0)
  end subroutine integrate_one

  ! use_env: RCONST holds the rate evaluator's inputs (MISTRA_RATES_ENV_x) instead of rate constants: Update_RCONST_x runs on the GPU too
  subroutine integrate_batch(mech, sfx, NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT, use_env)
    integer, intent(in) :: mech, NCELL
    character(len=1), intent(in) :: sfx
    logical, intent(in), optional :: use_env
    logical :: env
    real(c_double) :: VAR(*), FIX(*), RCONST(*), TEXIT(NCELL), HEXIT(NCELL)
    real(c_double), intent(in) :: TIN, TOUT
    integer(c_int32_t) :: IERR(NCELL), ISTAT(8, NCELL)
    real(c_double), allocatable :: th(:, :)
    integer :: k
    if (NCELL <= 0) return
    env = .false.
    if (present(use_env)) env = use_env
    allocate (th(3, NCELL))
    if (env) then
       if (mistra_chem_integrate_env_ex(int(mech, c_int), int(NCELL, c_int), VAR, FIX, RCONST, TIN, TOUT, VAR, IERR, ISTAT, th) /= 0) &
            call mistra_chem_fail('INTEGRATE_BATCH_ENV_'//sfx)
    else
       if (mistra_chem_integrate_ex(int(mech, c_int), int(NCELL, c_int), VAR, FIX, RCONST, TIN, TOUT, VAR, IERR, ISTAT, th) /= 0) &
            call mistra_chem_fail('INTEGRATE_BATCH_'//sfx)
    end if
    do k = 1, NCELL            ! the messages in layer order, as the serial loop would have written them
       TEXIT(k) = th(1, k)
       HEXIT(k) = th(2, k)
       if (IERR(k) < 0 .or. ISTAT(8, k) > 0) call mistra_chem_report(sfx, int(IERR(k)), th(1, k), th(3, k), TIN, int(ISTAT(8, k)), mech, k - 1)
    end do
    deallocate (th)
  end subroutine integrate_batch
end module mistra_chem_c_api

subroutine INTEGRATE_g(TIN, TOUT)
  use iso_c_binding
  use mistra_chem_c_api
  use mistra_kpp_batch
  implicit none
  real(c_double) :: TIN, TOUT
  integer, parameter :: NVAR = 102, NFIX = 3, NREACT = 331              ! gas_Parameters.h:28-49
  real(c_double), target :: C(NVAR + NFIX)
  real(c_double) :: RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
  common /GDATA_g/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  select case (kpp_pass)            ! two-pass layer loop of a batched kpp_driver (mistra_kpp_batch.f90); 0 = serial
  case (1)
     call kpp_batch_store(1, C, RCONST)
  case (2)
     call kpp_batch_fetch(1, C, TIN, STEPMIN)
     RTOL = 1.0d-3                  ! as INTEGRATE_x leaves them (gas.f:745-746)
     ATOL = 1.0d-25
  case default
     call integrate_one(0, 'g', c_loc(C), TIN, TOUT)
  end select
end subroutine INTEGRATE_g

subroutine INTEGRATE_a(TIN, TOUT)
  use iso_c_binding
  use mistra_chem_c_api
  use mistra_kpp_batch
  implicit none
  real(c_double) :: TIN, TOUT
  integer, parameter :: NVAR = 257, NFIX = 5, NREACT = 979              ! aer_Parameters.h:28-49
  real(c_double), target :: C(NVAR + NFIX)
  real(c_double) :: RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
  common /GDATA_a/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  select case (kpp_pass)            ! two-pass layer loop of a batched kpp_driver (mistra_kpp_batch.f90); 0 = serial
  case (1)
     call kpp_batch_store(2, C, RCONST)
  case (2)
     call kpp_batch_fetch(2, C, TIN, STEPMIN)
     RTOL = 1.0d-3                  ! as INTEGRATE_x leaves them (gas.f:745-746)
     ATOL = 1.0d-25
  case default
     call integrate_one(1, 'a', c_loc(C), TIN, TOUT)
  end select
end subroutine INTEGRATE_a

subroutine INTEGRATE_t(TIN, TOUT)
  use iso_c_binding
  use mistra_chem_c_api
  use mistra_kpp_batch
  implicit none
  real(c_double) :: TIN, TOUT
  integer, parameter :: NVAR = 417, NFIX = 7, NREACT = 1627             ! tot_Parameters.h:28-49
  real(c_double), target :: C(NVAR + NFIX)
  real(c_double) :: RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
  common /GDATA_t/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  select case (kpp_pass)            ! two-pass layer loop of a batched kpp_driver (mistra_kpp_batch.f90); 0 = serial
  case (1)
     call kpp_batch_store(3, C, RCONST)
  case (2)
     call kpp_batch_fetch(3, C, TIN, STEPMIN)
     RTOL = 1.0d-3                  ! as INTEGRATE_x leaves them (gas.f:745-746)
     ATOL = 1.0d-25
  case default
     call integrate_one(2, 't', c_loc(C), TIN, TOUT)
  end select
end subroutine INTEGRATE_t

! ---- batched form: what INTEGRATE_x does to COMMON /GDATA_x/, for NCELL cells at once.
!   VAR(NVAR,NCELL)  in: concentrations, out: after [TIN, TOUT]            (C(1:NVAR) of each cell)
!   FIX(NFIX,NCELL), RCONST(NREACT,NCELL)  in                              (C(NVAR+1:NSPEC), RCONST of each cell)
!   TEXIT(NCELL), HEXIT(NCELL)  out: what the serial call leaves in TIN and STEPMIN (gas.f:769-770)
!   IERR(NCELL)  out: 1 or the negative code of ros_ErrorMsg_x;  ISTAT(8,NCELL) out: COMMON /Statistics/ per cell
subroutine INTEGRATE_BATCH_g(NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  integer :: NCELL
  real(c_double) :: VAR(102, *), FIX(3, *), RCONST(331, *), TIN, TOUT, TEXIT(*), HEXIT(*)
  integer(c_int32_t) :: IERR(*), ISTAT(8, *)
  call integrate_batch(0, 'g', NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
end subroutine INTEGRATE_BATCH_g

subroutine INTEGRATE_BATCH_a(NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  integer :: NCELL
  real(c_double) :: VAR(257, *), FIX(5, *), RCONST(979, *), TIN, TOUT, TEXIT(*), HEXIT(*)
  integer(c_int32_t) :: IERR(*), ISTAT(8, *)
  call integrate_batch(1, 'a', NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
end subroutine INTEGRATE_BATCH_a

subroutine INTEGRATE_BATCH_t(NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  integer :: NCELL
  real(c_double) :: VAR(417, *), FIX(7, *), RCONST(1627, *), TIN, TOUT, TEXIT(*), HEXIT(*)
  integer(c_int32_t) :: IERR(*), ISTAT(8, *)
  call integrate_batch(2, 't', NCELL, VAR, FIX, RCONST, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
end subroutine INTEGRATE_BATCH_t

! ---- the same from the rate evaluator's inputs: Update_RCONST_x + INTEGRATE_x of NCELL layers in one call, RCONST never on the host.
!   ENV(nenv_x,NCELL): per layer what MISTRA_RATES_ENV_x (mistra_kpp_rates.f90) packs from the COMMON blocks
subroutine INTEGRATE_BATCH_ENV_g(NCELL, VAR, FIX, ENV, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  integer :: NCELL
  real(c_double) :: VAR(102, *), FIX(3, *), ENV(74, *), TIN, TOUT, TEXIT(*), HEXIT(*)
  integer(c_int32_t) :: IERR(*), ISTAT(8, *)
  call integrate_batch(0, 'g', NCELL, VAR, FIX, ENV, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT, .true.)
end subroutine INTEGRATE_BATCH_ENV_g

subroutine INTEGRATE_BATCH_ENV_a(NCELL, VAR, FIX, ENV, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  integer :: NCELL
  real(c_double) :: VAR(257, *), FIX(5, *), ENV(330, *), TIN, TOUT, TEXIT(*), HEXIT(*)
  integer(c_int32_t) :: IERR(*), ISTAT(8, *)
  call integrate_batch(1, 'a', NCELL, VAR, FIX, ENV, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT, .true.)
end subroutine INTEGRATE_BATCH_ENV_a

subroutine INTEGRATE_BATCH_ENV_t(NCELL, VAR, FIX, ENV, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  integer :: NCELL
  real(c_double) :: VAR(417, *), FIX(7, *), ENV(544, *), TIN, TOUT, TEXIT(*), HEXIT(*)
  integer(c_int32_t) :: IERR(*), ISTAT(8, *)
  call integrate_batch(2, 't', NCELL, VAR, FIX, ENV, TIN, TOUT, TEXIT, HEXIT, IERR, ISTAT, .true.)
end subroutine INTEGRATE_BATCH_ENV_t
