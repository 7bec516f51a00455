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
!          shim_driver D <in.bin> <out.bin>            one 10-s step of a column through the SINGLE-PASS batched driver (shim/mistra_kpp_drive.f90,
!                                                      shim/kpp_drive.patch): this program plays kpp_driver and x_drive's prologue — per layer it puts the
!                                                      recorded values into the COMMON blocks (shim_driver_env_set.f90), calls KPP_DRIVE_STAGE_x as the
!                                                      patched x_drive does, and behind the loop kpp_drive_run_arrays on its own copies of the model
!                                                      arrays (what KPP_DRIVE_RUN of mistra_kpp_model.f90 does inside the model).  in.bin (float64):
!                                                      n, j1, j5, nlev, nrxn, nl, nrep | maps of the three mechanisms | il(nlev) | s1(j1,n) s3(j5,n)
!                                                      sl1(121,4,n) sion1(55,4,n) bg(2,nrxn,nlev) bgs(2,122,n) | per layer: mech, k, air, h2o, env(nenv).
!                                                      out.bin: the arrays after the step | per layer ierr, 8 statistics, texit, hexit | per repetition
!                                                      the wall times (ms) of the staging loop and of the device call(s)
!          shim_driver <Ka|Kt|Ha|Ht|Va|Vt|Sa|St|Qa|Qt|Ca|Rg|Ra|Rt> <in.bin> <out.bin>   liq_parm's kernels through shim/mistra_kpp_liq.f90 (SURVEY §8 f3): K = FAST_K_MT_BATCH
!                                                      (in: nlayer, nka, nkt, nkc, nspec, ka, ifeed, nkc_l | kw | rq | ff cw cm freep alpha vmean xkmt t p vt;
!                                                      out: xkmt, vt), H = HENRY_BATCH (in: nlayer, nspec | tt; out: henry), V = V_MEAN_BATCH (the same shapes; out: vmean), S = ST_COEFF_BATCH (in: nlayer, nspec, lpJoyce14bc, lpBuxmann15alph | env(5,nlayer); out: alpha), R = DRY_RATES_BATCH (in: nlayer | tt, freep, rcd(2,nlayer), vmean4 / henry4 (4,nlayer); out: xkmtd, xeq, henry4), C = CW_RC_BATCH (in: nlayer, nkt, nka, dry, ka, ifeed | kw, rq, e, crys4, ff, feu, cloud; out: rc, cw, cm, conv2, below), Q = EQUIL_CO_BATCH (in: nlayer,
!                                                      nkc, j6, nspec | tt conv2 xgamma xkef xkeb; out: xkef, xkeb)
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
  case ('D'); call run_drive(trim(fin), trim(fout))
  case ('K', 'H', 'Q', 'V', 'S', 'C', 'R')
     select case (a1(2:2))
     case ('g')
        if (a1(1:1) /= 'R') stop 'mechanism must be a or t'
        call run_liq(a1(1:1), 1, trim(fin), trim(fout))
     case ('a'); call run_liq(a1(1:1), 2, trim(fin), trim(fout))
     case ('t'); call run_liq(a1(1:1), 3, trim(fin), trim(fout))
     case default; stop 'mechanism must be a or t'
     end select
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
  subroutine run_liq(what, mech, fin, fout)
    use mistra_kpp_liq
    character(len=1), intent(in) :: what
    integer, intent(in) :: mech
    character(len=*), intent(in) :: fin, fout
    double precision :: h(8)
    integer :: nl, nka, nkt, nkc, nspec, ka, ifeed, nkc_l, j6
    integer, allocatable :: kw(:)
    double precision, allocatable :: tmp(:), rq(:), ff(:), cw(:), cm(:), freep(:), alpha(:), vmean(:), xkmt(:), t(:), p(:), vt(:), henry(:), conv2(:), xgamma(:), &
                                     xkef(:), xkeb(:)
    open (11, file=fin, access='stream', form='unformatted', status='old')
    open (12, file=fout, access='stream', form='unformatted', status='replace')
    select case (what)
    case ('K')
       read (11) h
       nl = int(h(1)); nka = int(h(2)); nkt = int(h(3)); nkc = int(h(4)); nspec = int(h(5)); ka = int(h(6)); ifeed = int(h(7)); nkc_l = int(h(8))
       allocate (tmp(nka), kw(nka), rq(nkt * nka), ff(nkt * nka * nl), cw(nkc * nl), cm(nkc * nl), freep(nl), alpha(nspec * nl), vmean(nspec * nl), &
                 xkmt(nspec * nkc * nl), t(nl), p(nl), vt(nkc * nl))
       read (11) tmp; kw = int(tmp)
       read (11) rq, ff, cw, cm, freep, alpha, vmean, xkmt, t, p, vt
       call FAST_K_MT_BATCH(mech, nl, ff, rq, nka, kw, ka, ifeed, nkc_l, cw, cm, freep, alpha, vmean, xkmt, t, p, vt)
       write (12) xkmt, vt
    case ('H')
       read (11) h(1:2)
       nl = int(h(1)); nspec = int(h(2))
       allocate (t(nl), henry(nspec * nl))
       read (11) t
       henry = -7.d0
       call HENRY_BATCH(mech, nl, t, henry)
       write (12) henry
    case ('V')
       read (11) h(1:2)
       nl = int(h(1)); nspec = int(h(2))
       allocate (t(nl), vmean(nspec * nl))
       read (11) t
       vmean = -7.d0
       call V_MEAN_BATCH(mech, nl, t, vmean)
       write (12) vmean
    case ('S')
       read (11) h(1:4)
       nl = int(h(1)); nspec = int(h(2))
       allocate (t(5 * nl), alpha(nspec * nl))
       read (11) t
       alpha = -7.d0
       call ST_COEFF_BATCH(mech, nl, h(3) /= 0.d0, h(4) /= 0.d0, t, alpha)
       write (12) alpha
    case ('C')
       call run_cw_rc()
    case ('R')
       read (11) h(1:1)
       nl = int(h(1))
       allocate (t(nl), freep(nl), rq(2 * nl), vmean(4 * nl), xkmt(8 * nl), henry(nl), alpha(4 * nl))
       read (11) t, freep, rq, vmean
       alpha = vmean      ! (gas: the Henry entries before the call, in/out)
       xkmt = -7.d0; henry = -7.d0
       call DRY_RATES_BATCH(mech == 1, nl, t, freep, rq, vmean, xkmt, henry, alpha)
       write (12) xkmt, henry, alpha
    case ('Q')
       read (11) h(1:4)
       nl = int(h(1)); nkc = int(h(2)); j6 = int(h(3)); nspec = int(h(4))
       allocate (t(nl), conv2(nkc * nl), xgamma(j6 * nkc * nl), xkef(nspec * nkc * nl), xkeb(nspec * nkc * nl))
       read (11) t, conv2, xgamma, xkef, xkeb
       call EQUIL_CO_BATCH(mech, nl, nkc, j6, t, conv2, xgamma, xkef, xkeb)
       write (12) xkef, xkeb
    end select
    close (11); close (12)
  end subroutine run_liq
  subroutine run_cw_rc()      ! (units 11 and 12 are open)
    use mistra_kpp_liq
    double precision :: h(6), crys4(4)
    integer :: nl, nkt, nka, ka, ifeed, nb
    logical :: dry
    integer, allocatable :: kw(:), cloud(:), below(:)
    double precision, allocatable :: tmp(:), rq(:), e(:), ff(:), feu(:), rc(:), cw(:), cm(:), conv2(:)
    read (11) h
    nl = int(h(1)); nkt = int(h(2)); nka = int(h(3)); dry = h(4) /= 0.d0; ka = int(h(5)); ifeed = int(h(6))
    nb = merge(2, 4, dry)
    allocate (tmp(nka), kw(nka), rq(nkt * nka), e(nkt), ff(nkt * nka * nl), feu(nl), cloud(4 * nl), below(nl), rc(nb * nl), cw(nb * nl), cm(nb * nl), conv2(nb * nl))
    read (11) tmp; kw = int(tmp)
    read (11) rq, e, crys4, ff, feu
    deallocate (tmp); allocate (tmp(4 * nl))
    read (11) tmp; cloud = int(tmp)
    rc = -7.d0; cw = -7.d0; cm = -7.d0; conv2 = -7.d0; below = -7
    call CW_RC_BATCH(nl, nkt, nka, dry, ff, rq, e, kw, ka, ifeed, feu, cloud, crys4, rc, cw, cm, conv2, below)
    write (12) rc, cw, cm, conv2, dble(below)
    ! once more with the spectrum registered for direct transfers, as LIQ_PIN_ONCE does for the model's /cb52/ (same bits; the test compares)
    call PIN_HOST(ff, nkt * nka * nl)
    rc = -7.d0; cw = -7.d0; cm = -7.d0; conv2 = -7.d0; below = -7
    call CW_RC_BATCH(nl, nkt, nka, dry, ff, rq, e, kw, ka, ifeed, feu, cloud, crys4, rc, cw, cm, conv2, below)
    write (12) rc, cw, cm, conv2, dble(below)
  end subroutine run_cw_rc
  subroutine run_drive(fin, fout)
    use mistra_kpp_drive
    character(len=*), intent(in) :: fin, fout
    integer, parameter :: j2 = 121, j6 = 55, nkc = 4, nbgs = 122      ! global_params.f90:96-103, bud_s_g.f:63
    integer, parameter :: nenv(3) = [nenv_g, nenv_a, nenv_t]
    double precision :: hdr(7)
    integer :: n, j1, j5, nlev, nrxn, nl, nrep, i, m, rep, cnt, k, ierr, istat(8)
    integer, allocatable :: gm(:, :, :), gk(:, :), rm(:, :, :), rk(:, :), il(:), lmech(:), lk(:)
    double precision, allocatable :: tmp(:), s1(:, :), s3(:, :), sl1(:, :, :), sion1(:, :, :), bg(:, :, :), bgs(:, :, :), lscal(:, :), lenv(:, :)
    double precision, allocatable :: s1_0(:, :), s3_0(:, :), sl1_0(:, :, :), sion1_0(:, :, :), bg_0(:, :, :), bgs_0(:, :, :), times(:, :)
    double precision :: texit, hexit
    integer(8) :: c0, c1, c2, rate
    external :: MISTRA_RATES_ENV_SET_g, MISTRA_RATES_ENV_SET_a, MISTRA_RATES_ENV_SET_t, KPP_DRIVE_STAGE_g, KPP_DRIVE_STAGE_a, KPP_DRIVE_STAGE_t
    open (11, file=fin, access='stream', form='unformatted', status='old')
    open (12, file=fout, access='stream', form='unformatted', status='replace')
    read (11) hdr
    n = int(hdr(1)); j1 = int(hdr(2)); j5 = int(hdr(3)); nlev = int(hdr(4)); nrxn = int(hdr(5)); nl = int(hdr(6)); nrep = int(hdr(7))
    allocate (gm(2, j1, 3), gk(j1, 3), rm(2, j5, 3), rk(j5, 3), il(nlev), tmp(max(2 * j1, 2 * j5, nlev)))
    do m = 1, 3
       read (11) tmp(1:2 * j1); gm(:, :, m) = reshape(int(tmp(1:2 * j1)), [2, j1])
       read (11) tmp(1:j1); gk(:, m) = int(tmp(1:j1))
       read (11) tmp(1:2 * j5); rm(:, :, m) = reshape(int(tmp(1:2 * j5)), [2, j5])
       read (11) tmp(1:j5); rk(:, m) = int(tmp(1:j5))
    end do
    read (11) tmp(1:nlev); il = int(tmp(1:nlev))
    allocate (s1(j1, n), s3(j5, n), sl1(j2, nkc, n), sion1(j6, nkc, n), bg(2, nrxn, nlev), bgs(2, nbgs, n))
    read (11) s1, s3, sl1, sion1, bg, bgs
    allocate (lmech(nl), lk(nl), lscal(2, nl), lenv(maxval(nenv), nl), times(2, nrep))
    do i = 1, nl
       read (11) hdr(1:4)
       lmech(i) = int(hdr(1)); lk(i) = int(hdr(2)); lscal(:, i) = hdr(3:4)
       read (11) lenv(1:nenv(lmech(i)), i)
    end do
    s1_0 = s1; s3_0 = s3; sl1_0 = sl1; sion1_0 = sion1; bg_0 = bg; bgs_0 = bgs
    do m = 1, 3      ! once per run, as KPP_DRIVE_RUN does on its first call (here only for the mechanisms the recorded step has layers of: the
       if (any(lmech == m)) call kpp_drive_maps(m, j1, gm(:, :, m), gk(:, m), j5, rm(:, :, m), rk(:, m))      ! file holds real maps only for those)
    end do
    do rep = 1, nrep      ! (the first repetition pays the library's start-up and the first allocation of the staging blocks)
       s1 = s1_0; s3 = s3_0; sl1 = sl1_0; sion1 = sion1_0; bg = bg_0; bgs = bgs_0
       call system_clock(c0, rate)
       call kpp_drive_begin
       do i = 1, nl       ! kpp_driver's layer loop: per-layer set-up and the x_drive prologue (here: the recorded values back into the COMMON blocks), then the hand-over
          select case (lmech(i))
          case (1); call MISTRA_RATES_ENV_SET_g(lenv(:, i)); call KPP_DRIVE_STAGE_g(lk(i), lscal(1, i), lscal(2, i))
          case (2); call MISTRA_RATES_ENV_SET_a(lenv(:, i)); call KPP_DRIVE_STAGE_a(lk(i), lscal(1, i), lscal(2, i))
          case (3); call MISTRA_RATES_ENV_SET_t(lenv(:, i)); call KPP_DRIVE_STAGE_t(lk(i), lscal(1, i), lscal(2, i))
          end select
       end do
       call system_clock(c1)
       call kpp_drive_run_arrays(0.d0, 10.d0, n, s1, s3, sl1, sion1, nrxn, nlev, il, bg, bgs)
       call system_clock(c2)
       times(1, rep) = 1.d3 * dble(c1 - c0) / dble(rate)
       times(2, rep) = 1.d3 * dble(c2 - c1) / dble(rate)
    end do
    write (12) s1, s3, sl1, sion1, bg, bgs
    do m = 1, 3           ! per mechanism in staging order (= layer order within the mechanism)
       cnt = kpp_drive_count(m)
       do i = 1, cnt
          call kpp_drive_last(m, i, k, ierr, istat, texit, hexit)
          write (12) dble(m), dble(k), dble(ierr), dble(istat), texit, hexit
       end do
    end do
    write (12) times
    close (11); close (12)
  end subroutine run_drive
end program shim_driver
