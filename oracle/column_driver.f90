! TEST INFRASTRUCTURE (fixture capture only; built by oracle/build_ref.sh `model`, never shipped, never timed).
!
! A stripped time loop that sequences the reference's own, unmodified, compiled routines for TWO configuration
! classes: 1-D column, or box (box=T: one level, src/str.f90:183-186, 215, 305, 410-424) — microphysics on, chemistry on,
! with or without the nucleation module (src/str.f90:206-210, 397-402), water surface, fresh start.  It exists because
! the reference's main program (src/str.f90:72-560) calls write_grid (src/str.f90:216) unconditionally, and that
! routine lives in src/out_netCDF.f which needs netcdf.inc/libnetcdf — absent from this image.  Instead of faking
! that library, this harness simply never calls any output routine (netCDF, binary plots, restart files): those
! do not feed back into the model state.  Everything that does (dynamics, microphysics, radiation, photolysis,
! chemistry stem) is called in the order the reference's loop uses (src/str.f90:198-300 start-up, 324-516 loop).
!
! Purpose: reach realistic cloudy-layer chemistry states so that oracle/capture_wrap.c can record real
! INTEGRATE_g/a/t inputs and outputs (/GDATA_x/) for tests/golden/.
program mistra_column_capture
  use config, only: read_config, box, chamber, chem, mic, nuc, rst, isurf, lstmax, z_box, lpJoyce14bc, lpBuys13_0D, nlevbox, BL_box, &
                    Napari, Lovejoy, iod
  use global_params, only: n, nf, nm, nphrxn, nrlay, mbs
  use precision, only: dp
  implicit none

  interface
     subroutine equil(ncase, kk)
       integer, intent(in) :: ncase
       integer, optional, intent(in) :: kk
     end subroutine equil
  end interface

  real(dp), parameter :: dt_slow = 60._dp, dt_fast = 10._dp
  character(len=1), parameter :: tag = 'a'
  integer :: minutes, sub, k, nbl, max_minutes, envstat, nz_box
  character(len=32) :: envbuf
  real(dp) :: xra, u0_floor, box_switch
  logical :: daylight, llnucboth
  ! end-to-end runs (oracle/build_gpu_model.sh, tests/test_gpu_model.py): wall time of the chemistry stem, the model's chemical state at the end
  integer(8) :: clk0, clk1, clk_rate, clk_chem, clk_start
  character(len=512) :: dumpfile
  integer :: dumpstat

  ! the handful of reference COMMON members this loop has to advance itself
  real(dp) :: u0, albedo, thk
  common /cb16/ u0, albedo(mbs), thk(nrlay)
  real(dp) :: time
  integer :: lday, lst, lmin, it, lcl, lct
  common /cb40/ time, lday, lst, lmin, it, lcl, lct
  real(dp) :: sk, sl, dtrad, dtcon
  common /cb48/ sk, sl, dtrad(n), dtcon(n)
  real(dp) :: theta, thetl, t, talt, p, rho
  common /cb53/ theta(n), thetl(n), t(n), talt(n), p(n), rho(n)
  real(dp) :: photol_j
  common /band_rat/ photol_j(nphrxn, n)

  call read_config
  if (chamber .or. rst .or. (.not. chem) .or. (.not. mic) .or. isurf /= 0) then
     write (0, *) 'mistra_column_capture: only 1-D or box, mic=T, chem=T, rst=F, isurf=0 namelists are supported'
     stop 2
  end if
  nbl = merge(2, nf, box)                 ! src/str.f90:183-196
  xra = 0._dp
  nz_box = 0

  ! ---- start-up
  call grid
  call openm(tag)
  call openc(tag)
  call mk_interface
  llnucboth = .false.
  if (nuc) then                           ! src/str.f90:206-210
     call nuc_init(Napari, Lovejoy, iod)
     llnucboth = Napari .and. Lovejoy
  end if
  if (box) call get_n_box(z_box, nz_box)   ! src/str.f90:215
  call initm(tag, rst)
  call initc(nbl)
  call atk0
  call equil(0)
  call radiation(.true.)
  call oneD_dist_jjb
  call constm
  call constc
  call profm(dt_slow)
  call profc(dt_slow, mic)
  call photol_initialize
  call photol
  call out_mass
  time = 0._dp
  if (box) then                           ! src/str.f90:305-306
     call box_init(nlevbox, nz_box, nbl, BL_box)
     box_switch = 1._dp
  end if
  u0_floor = merge(1.75e-2_dp, 3.48e-2_dp, lpBuys13_0D)

  ! ---- minutes
  call get_environment_variable('MISTRA_COLUMN_MINUTES', envbuf, status=envstat)      ! shorter runs for the test-suite
  max_minutes = 60 * lstmax
  if (envstat == 0) read (envbuf, *) max_minutes
  clk_chem = 0
  call system_clock(clk_start, clk_rate)
  do minutes = 1, min(max_minutes, 60 * lstmax)
     it = minutes
     if (lct > nf) stop 'cloud top above nf'
     lmin = lmin + 1
     if (lmin == 60) then
        lmin = 0
        lst = lst + 1
        if (lst == 24) then
           lst = 0
           lday = lday + 1
        end if
     end if
     call partdep(xra)
     do sub = 1, 6
        time = time + dt_fast
        if (box) then                     ! src/str.f90:410-424: no dynamics, microphysics or radiation in a box run
           call box_update(box_switch, sub, nlevbox, nz_box, nbl, BL_box)
           call sedc_box(dt_fast, z_box, nbl)
           call box_partdep(dt_fast, z_box, nbl)
           call system_clock(clk0)
           call stem_kpp(dt_fast, xra, z_box, nbl, box, chamber, nuc)
           call system_clock(clk1); clk_chem = clk_chem + (clk1 - clk0)
           cycle
        end if
        call difm(dt_fast)
        call difc(dt_fast)
        call difp(dt_fast)
        call kon(dt_fast, chem)
        call sedp(dt_fast)
        call equil(2)
        do k = 2, nm
           t(k) = t(k) + dtrad(k) * dt_fast
        end do
        call surf0(dt_fast)
        call sedc(dt_fast)
        call sedl(dt_fast)
        call system_clock(clk0)
        call stem_kpp(dt_fast, xra, z_box, nbl, box, chamber, nuc)
        call system_clock(clk1); clk_chem = clk_chem + (clk1 - clk0)
        if (nuc) then                     ! src/str.f90:397-402
           if (llnucboth) then
              call appnucl2(dt_fast, llnucboth)
           else
              call appnucl(dt_fast, Napari, Lovejoy, llnucboth)
           end if
        end if
     end do
     if (.not. box) call radiation(.false.)
     ! photolysis refresh rule of the reference loop
     if (lpJoyce14bc) then
        daylight = u0 > 1.0e-2_dp
        if (daylight) then
           call photol
        else
           photol_j(:, :) = 0._dp
        end if
     else if (u0 > u0_floor) then
        if (mod(lmin, 2) == 0) then
           call photol
           if (box .and. BL_box) call ave_j(nz_box, nbl)
        end if
     else
        photol_j(:, :) = 0._dp
     end if
     if (mod(lmin, 15) == 0) then
        call oneD_dist_jjb
        call out_mass
     end if
     if (mod(lmin, 60) == 0) then
        call profm(dt_slow)
        call profc(dt_slow, mic)
     end if
     if (mod(minutes, 10) == 0) write (0, '(a,i5,a,i3,a,i3)') ' [column] minute ', minutes, '  lcl ', lcl, '  lct ', lct
  end do
  call system_clock(clk1)
  write (0, '(a,i6,a,f10.3,a,f10.3,a)') ' [column] ', min(max_minutes, 60 * lstmax) * 6, ' steps of 10 s: chemistry stem (liq_parm + kpp_driver + the rest of stem_kpp) ', &
       dble(clk_chem) / dble(clk_rate), ' s of ', dble(clk1 - clk_start) / dble(clk_rate), ' s in the time loop'
  call get_environment_variable('MISTRA_COLUMN_DUMP', dumpfile, status=dumpstat)
  if (dumpstat == 0 .and. len_trim(dumpfile) > 0) call dump_chemical_state(trim(dumpfile))
contains
  ! the chemical state the column ends in: s1, s3 of module gas_common, sl1 / sion1 of /blck17/, the temperature profile; stream of doubles behind
  ! four int32 {j1, j5, j2*nkc, j6*nkc} and n
  subroutine dump_chemical_state(path)
    use gas_common, only: j1, j5, s1, s3
    use global_params, only: j2, j6, nkc
    character(len=*), intent(in) :: path
    real(dp) :: sl1, sion1
    common /blck17/ sl1(j2, nkc, n), sion1(j6, nkc, n)
    integer :: u
    open (newunit=u, file=path, access='stream', form='unformatted', status='replace')
    write (u) int(j1, 4), int(j5, 4), int(j2 * nkc, 4), int(j6 * nkc, 4), int(n, 4)
    write (u) s1(1:j1, 1:n), s3(1:j5, 1:n), sl1, sion1, t
    close (u)
  end subroutine dump_chemical_state
end program mistra_column_capture
