! Two-pass layer loop for kpp_driver (kpp.f90:4310-4470): all layers of a 10-s step that run the same mechanism go to the
! GPU as ONE batch (INTEGRATE_BATCH_x, mistra_kpp_shim.f90) instead of one INTEGRATE_x call per layer.
!
! The reference's loop calls x_drive(k) per layer; x_drive packs COMMON /GDATA_x/ from the model arrays, calls
! Update_RCONST_x and INTEGRATE_x, then runs the budgets and hands the concentrations back (gas.f:60-270 | aer.f:59-330 |
! tot.f:59-982).  With the patch of INTEGRATION.md §4 kpp_driver runs its loop twice:
!   pass 1  kpp_pass = 1: x_drive packs and computes the rates as always; INTEGRATE_x (this shim's) only RECORDS the
!           layer's C and RCONST and returns; x_drive returns right behind it (the one line the patch adds to each x_drive)
!   between kpp_batch_run: one INTEGRATE_BATCH_x call per mechanism that has layers in this step
!   pass 2  kpp_pass = 2: x_drive packs and computes the rates again (same inputs, same values), INTEGRATE_x hands back
!           the layer's integrated C, exit time and last step from the batch, and x_drive goes on into its budget and
!           hand-over half with exactly the COMMON block contents the serial call would have left
! kpp_pass = 0 (default) is the serial path: one GPU call per INTEGRATE_x.
module mistra_kpp_batch
  use iso_c_binding
  implicit none
  integer :: kpp_pass = 0
  integer, parameter :: nmech = 3
  integer, parameter :: mvar(nmech) = [102, 257, 417], mfix(nmech) = [3, 5, 7], mreact(nmech) = [331, 979, 1627]
  type layer_batch
     integer :: n = 0, next = 0                        ! layers recorded in pass 1 / handed back so far in pass 2
     real(c_double), allocatable :: var(:, :), fix(:, :), rconst(:, :), texit(:), hexit(:)
     integer(c_int32_t), allocatable :: ierr(:), istat(:, :)
  end type layer_batch
  type(layer_batch), save :: batch(nmech)
contains
  subroutine kpp_batch_begin()
    integer :: m
    do m = 1, nmech
       batch(m)%n = 0
       batch(m)%next = 0
    end do
  end subroutine kpp_batch_begin

  subroutine grow(b, m)
    type(layer_batch), intent(inout) :: b
    integer, intent(in) :: m
    type(layer_batch) :: t
    integer :: cap
    cap = 0
    if (allocated(b%var)) cap = size(b%var, 2)
    if (b%n < cap) return
    cap = max(64, 2 * cap)
    allocate (t%var(mvar(m), cap), t%fix(mfix(m), cap), t%rconst(mreact(m), cap), t%texit(cap), t%hexit(cap), t%ierr(cap), t%istat(8, cap))
    if (b%n > 0) then
       t%var(:, 1:b%n) = b%var(:, 1:b%n)
       t%fix(:, 1:b%n) = b%fix(:, 1:b%n)
       t%rconst(:, 1:b%n) = b%rconst(:, 1:b%n)
    end if
    call move_alloc(t%var, b%var); call move_alloc(t%fix, b%fix); call move_alloc(t%rconst, b%rconst)
    call move_alloc(t%texit, b%texit); call move_alloc(t%hexit, b%hexit); call move_alloc(t%ierr, b%ierr); call move_alloc(t%istat, b%istat)
  end subroutine grow

  ! pass 1, called by INTEGRATE_x: C = VAR | FIX and RCONST of this layer as x_drive and Update_RCONST_x have just set them
  subroutine kpp_batch_store(m, C, RCONST)
    integer, intent(in) :: m
    real(c_double), intent(in) :: C(*), RCONST(*)
    call grow(batch(m), m)
    batch(m)%n = batch(m)%n + 1
    batch(m)%var(:, batch(m)%n) = C(1:mvar(m))
    batch(m)%fix(:, batch(m)%n) = C(mvar(m) + 1:mvar(m) + mfix(m))
    batch(m)%rconst(:, batch(m)%n) = RCONST(1:mreact(m))
  end subroutine kpp_batch_store

  ! between the passes: every mechanism's layers in one call (the messages of failed layers come out in layer order)
  subroutine kpp_batch_run(tin, tout)
    real(c_double), intent(in) :: tin, tout
    real(c_double) :: t0, t1
    integer :: m
    external :: INTEGRATE_BATCH_g, INTEGRATE_BATCH_a, INTEGRATE_BATCH_t
    do m = 1, nmech
       if (batch(m)%n == 0) cycle
       t0 = tin
       t1 = tout
       select case (m)
       case (1); call INTEGRATE_BATCH_g(batch(m)%n, batch(m)%var, batch(m)%fix, batch(m)%rconst, t0, t1, batch(m)%texit, batch(m)%hexit, batch(m)%ierr, batch(m)%istat)
       case (2); call INTEGRATE_BATCH_a(batch(m)%n, batch(m)%var, batch(m)%fix, batch(m)%rconst, t0, t1, batch(m)%texit, batch(m)%hexit, batch(m)%ierr, batch(m)%istat)
       case (3); call INTEGRATE_BATCH_t(batch(m)%n, batch(m)%var, batch(m)%fix, batch(m)%rconst, t0, t1, batch(m)%texit, batch(m)%hexit, batch(m)%ierr, batch(m)%istat)
       end select
    end do
  end subroutine kpp_batch_run

  ! pass 2, called by INTEGRATE_x: the layers come back in the order pass 1 recorded them (same loop, same order)
  subroutine kpp_batch_fetch(m, VAR, TIN, STEPMIN)
    integer, intent(in) :: m
    real(c_double), intent(out) :: VAR(*), TIN, STEPMIN
    integer :: i
    i = batch(m)%next + 1
    if (i > batch(m)%n) error stop 'mistra_kpp_batch: pass 2 asks for more layers than pass 1 recorded'
    batch(m)%next = i
    VAR(1:mvar(m)) = batch(m)%var(:, i)
    TIN = batch(m)%texit(i)
    STEPMIN = batch(m)%hexit(i)
  end subroutine kpp_batch_fetch
end module mistra_kpp_batch
