! Test driver for the Fortran shim (tests/test_gpu_shim.py): plays the part of x_drive — fills COMMON /GDATA_x/ from a
! raw file, calls INTEGRATE_x(TIN, TOUT) exactly as gas.f:173 | aer.f:217 | tot.f:604 do, writes the COMMON block back.
!   usage: shim_driver <g|a|t> <in.bin> <out.bin>      in.bin = ncell, then per cell C(NSPEC), RCONST(NREACT)  (float64)
program shim_driver
  implicit none
  character(len=256) :: a1, fin, fout
  call get_command_argument(1, a1)
  call get_command_argument(2, fin)
  call get_command_argument(3, fout)
  select case (a1(1:1))
  case ('g'); call run_g(trim(fin), trim(fout))
  case ('a'); call run_a(trim(fin), trim(fout))
  case ('t'); call run_t(trim(fin), trim(fout))
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
       write (12) C(1:NVAR), tkpp, STEPMIN
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
       write (12) C(1:NVAR), tkpp, STEPMIN
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
       write (12) C(1:NVAR), tkpp, STEPMIN
    end do
    close (11); close (12)
  end subroutine run_t
end program shim_driver
