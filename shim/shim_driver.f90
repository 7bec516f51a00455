! Test driver for the Fortran shim (tests/test_gpu_shim.py): plays the part of x_drive — fills COMMON /GDATA_x/ from a
! raw file, calls INTEGRATE_x(TIN, TOUT) exactly as gas.f:173 | aer.f:217 | tot.f:604 do, writes the COMMON block back.
!   usage: shim_driver <g|a|t> <in.bin> <out.bin>      in.bin = ncell, then per cell C(NSPEC), RCONST(NREACT)  (float64)
!          shim_driver <G|A|T> <in.bin> <out.bin>      the same cells as ONE batched call, INTEGRATE_BATCH_x — the call a two-pass
!                                                      kpp_driver makes per mechanism and 10-s step (INTEGRATION.md); out.bin then
!                                                      ends with per cell IERR and the 8 statistics, and the call's wall time in ms
!          shim_driver <Eg|Ea|Et> <in.bin> <out.bin>   in.bin = ncell, then per cell VAR, FIX, ENV (the vector MISTRA_RATES_ENV_x packs,
!                                                      mistra_kpp_rates.f90): UPDATE_RCONST_BATCH_x, then INTEGRATE_BATCH_ENV_x (rates and
!                                                      integrator on the device, RCONST never crosses PCIe); out.bin = per cell VAR,
!                                                      then per cell IERR + 8 statistics, then per cell RCONST of the first call
! After the call the one-cell mode also writes ATOL(1), RTOL(1) (INTEGRATE_x resets them, gas.f:745-746).
program shim_driver
  use mistra_kpp_rates
  implicit none
  character(len=256) :: a1, fin, fout
  call get_command_argument(1, a1)
  call get_command_argument(2, fin)
  call get_command_argument(3, fout)
  select case (a1(1:1))
  case ('g'); call run_g(trim(fin), trim(fout))
  case ('a'); call run_a(trim(fin), trim(fout))
  case ('t'); call run_t(trim(fin), trim(fout))
  case ('G'); call run_batch(0, 102, 3, 331, trim(fin), trim(fout))
  case ('A'); call run_batch(1, 257, 5, 979, trim(fin), trim(fout))
  case ('T'); call run_batch(2, 417, 7, 1627, trim(fin), trim(fout))
  case ('E')
     select case (a1(2:2))
     case ('g'); call run_env(0, 102, 3, 331, nenv_g, trim(fin), trim(fout))
     case ('a'); call run_env(1, 257, 5, 979, nenv_a, trim(fin), trim(fout))
     case ('t'); call run_env(2, 417, 7, 1627, nenv_t, trim(fin), trim(fout))
     case default; stop 'mechanism must be g, a or t'
     end select
  case default; stop 'mechanism must be g, a or t'
  end select
contains
  subroutine run_g(fin, fout)
    character(len=*), intent(in) :: fin, fout
    integer, parameter :: NVAR = 102, NFIX = 3, NREACT = 331
    double precision :: C(NVAR + NFIX), RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
    common /GDATA_g/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
    double precision :: tkpp, tend, rn
    integer :: n, i
    open (11, file=fin, access='stream', form='unformatted', status='old')
    open (12, file=fout, access='stream', form='unformatted', status='replace')
    read (11) rn
    n = int(rn)
    do i = 1, n
       read (11) C, RCONST
       tkpp = 0.d0
       tend = 10.d0
       call INTEGRATE_g(tkpp, tend)
       write (12) C(1:NVAR), tkpp, STEPMIN, ATOL(1), RTOL(1)
    end do
    close (11); close (12)
  end subroutine run_g
  subroutine run_a(fin, fout)
    character(len=*), intent(in) :: fin, fout
    integer, parameter :: NVAR = 257, NFIX = 5, NREACT = 979
    double precision :: C(NVAR + NFIX), RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
    common /GDATA_a/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
    double precision :: tkpp, tend, rn
    integer :: n, i
    open (11, file=fin, access='stream', form='unformatted', status='old')
    open (12, file=fout, access='stream', form='unformatted', status='replace')
    read (11) rn
    n = int(rn)
    do i = 1, n
       read (11) C, RCONST
       tkpp = 0.d0
       tend = 10.d0
       call INTEGRATE_a(tkpp, tend)
       write (12) C(1:NVAR), tkpp, STEPMIN, ATOL(1), RTOL(1)
    end do
    close (11); close (12)
  end subroutine run_a
  subroutine run_t(fin, fout)
    character(len=*), intent(in) :: fin, fout
    integer, parameter :: NVAR = 417, NFIX = 7, NREACT = 1627
    double precision :: C(NVAR + NFIX), RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
    common /GDATA_t/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
    double precision :: tkpp, tend, rn
    integer :: n, i
    open (11, file=fin, access='stream', form='unformatted', status='old')
    open (12, file=fout, access='stream', form='unformatted', status='replace')
    read (11) rn
    n = int(rn)
    do i = 1, n
       read (11) C, RCONST
       tkpp = 0.d0
       tend = 10.d0
       call INTEGRATE_t(tkpp, tend)
       write (12) C(1:NVAR), tkpp, STEPMIN, ATOL(1), RTOL(1)
    end do
    close (11); close (12)
  end subroutine run_t
  subroutine run_batch(mech, NVAR, NFIX, NREACT, fin, fout)
    integer, intent(in) :: mech, NVAR, NFIX, NREACT
    character(len=*), intent(in) :: fin, fout
    double precision, allocatable :: VAR(:, :), VAR0(:, :), FIX(:, :), RCONST(:, :), TEXIT(:), HEXIT(:), rec(:)
    integer, allocatable :: IERR(:), ISTAT(:, :)
    double precision :: rn, tin, tout
    integer :: n, i, rep
    integer(8) :: c0, c1, rate
    open (11, file=fin, access='stream', form='unformatted', status='old')
    open (12, file=fout, access='stream', form='unformatted', status='replace')
    read (11) rn
    n = int(rn)
    allocate (VAR(NVAR, n), VAR0(NVAR, n), FIX(NFIX, n), RCONST(NREACT, n), TEXIT(n), HEXIT(n), IERR(n), ISTAT(8, n), rec(NVAR + NFIX + NREACT))
    do i = 1, n                      ! pass 1 of a batched kpp_driver: every layer's C and RCONST, as x_drive prepares them
       read (11) rec
       VAR0(:, i) = rec(1:NVAR)
       FIX(:, i) = rec(NVAR + 1:NVAR + NFIX)
       RCONST(:, i) = rec(NVAR + NFIX + 1:)
    end do
    do rep = 1, 2                    ! the second call is the timed one (the first pays the library's start-up)
       VAR = VAR0
       tin = 0.d0
       tout = 10.d0
       call system_clock(c0, rate)
       select case (mech)
       case (0); call INTEGRATE_BATCH_g(n, VAR, FIX, RCONST, tin, tout, TEXIT, HEXIT, IERR, ISTAT)
       case (1); call INTEGRATE_BATCH_a(n, VAR, FIX, RCONST, tin, tout, TEXIT, HEXIT, IERR, ISTAT)
       case (2); call INTEGRATE_BATCH_t(n, VAR, FIX, RCONST, tin, tout, TEXIT, HEXIT, IERR, ISTAT)
       end select
       call system_clock(c1)
    end do
    do i = 1, n
       write (12) VAR(:, i), TEXIT(i), HEXIT(i), 1.d-25, 1.d-3
    end do
    do i = 1, n
       write (12) dble(IERR(i)), dble(ISTAT(:, i))
    end do
    write (12) 1.d3 * dble(c1 - c0) / dble(rate)
    close (11); close (12)
  end subroutine run_batch
  subroutine run_env(mech, NVAR, NFIX, NREACT, NENV, fin, fout)
    integer, intent(in) :: mech, NVAR, NFIX, NREACT, NENV
    character(len=*), intent(in) :: fin, fout
    double precision, allocatable :: VAR(:, :), FIX(:, :), ENV(:, :), RCONST(:, :), TEXIT(:), HEXIT(:), rec(:)
    integer, allocatable :: IERR(:), ISTAT(:, :)
    double precision :: rn, tin, tout
    integer :: n, i
    open (11, file=fin, access='stream', form='unformatted', status='old')
    open (12, file=fout, access='stream', form='unformatted', status='replace')
    read (11) rn
    n = int(rn)
    allocate (VAR(NVAR, n), FIX(NFIX, n), ENV(NENV, n), RCONST(NREACT, n), TEXIT(n), HEXIT(n), IERR(n), ISTAT(8, n), rec(NVAR + NFIX + NENV))
    do i = 1, n
       read (11) rec
       VAR(:, i) = rec(1:NVAR)
       FIX(:, i) = rec(NVAR + 1:NVAR + NFIX)
       ENV(:, i) = rec(NVAR + NFIX + 1:)
    end do
    tin = 0.d0
    tout = 10.d0
    select case (mech)
    case (0)
       call UPDATE_RCONST_BATCH_g(n, ENV, RCONST)
       call INTEGRATE_BATCH_ENV_g(n, VAR, FIX, ENV, tin, tout, TEXIT, HEXIT, IERR, ISTAT)
    case (1)
       call UPDATE_RCONST_BATCH_a(n, ENV, RCONST)
       call INTEGRATE_BATCH_ENV_a(n, VAR, FIX, ENV, tin, tout, TEXIT, HEXIT, IERR, ISTAT)
    case (2)
       call UPDATE_RCONST_BATCH_t(n, ENV, RCONST)
       call INTEGRATE_BATCH_ENV_t(n, VAR, FIX, ENV, tin, tout, TEXIT, HEXIT, IERR, ISTAT)
    end select
    do i = 1, n
       write (12) VAR(:, i)
    end do
    do i = 1, n
       write (12) dble(IERR(i)), dble(ISTAT(:, i))
    end do
    do i = 1, n
       write (12) RCONST(:, i)
    end do
    close (11); close (12)
  end subroutine run_env
end program shim_driver
