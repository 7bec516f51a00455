! Fortran side of the drop-in boundary: the reference's own subroutine surface,
!     SUBROUTINE INTEGRATE_g / INTEGRATE_a / INTEGRATE_t (TIN, TOUT)          gas.f:710 | aer.f:1408 | tot.f:2812
! implemented on top of the C ABI (include/mistra_chem.h) through ISO_C_BINDING.
!
! Same names, same two REAL*8 arguments by reference, same data path: everything else travels through
! COMMON /GDATA_x/, whose member order is restated below from gas_Global.h:29-58 (aer_Global.h, tot_Global.h alike):
!     C(NSPEC) [= VAR(NVAR) followed by FIX(NFIX)], RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
! A maintainer drops this file into src/ and removes (or renames) the three generated INTEGRATE_x routines; x_drive,
! Update_RCONST_x, the budgets and kpp_driver stay untouched (INTEGRATION.md shows the link line and the
! link-time alternative that needs no source edit).  Behaviour kept from the reference: VAR is advanced in place,
! TIN returns the exit time, STEPMIN the last step, RTOL/ATOL are (re)set to 1e-3 / 1e-25, an unsuccessful
! integration prints a message and the model carries on (gas.f:764-767).
module mistra_chem_c_api
  use iso_c_binding
  implicit none
  interface
     function mistra_chem_integrate_common(mech, gdata, tin, tout) bind(C, name="mistra_chem_integrate_common") result(rc)
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: mech
       type(c_ptr), value :: gdata
       real(c_double) :: tin, tout
       integer(c_int) :: rc
     end function mistra_chem_integrate_common
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
    stop 'mistra_chem: GPU integrator unavailable (there is no CPU fallback)'
  end subroutine mistra_chem_fail
end module mistra_chem_c_api

subroutine INTEGRATE_g(TIN, TOUT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  real(c_double) :: TIN, TOUT
  integer, parameter :: NVAR = 102, NFIX = 3, NREACT = 331              ! gas_Parameters.h:28-49
  real(c_double), target :: C(NVAR + NFIX)
  real(c_double) :: RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
  common /GDATA_g/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  if (mistra_chem_integrate_common(0_c_int, c_loc(C), TIN, TOUT) /= 0) call mistra_chem_fail('INTEGRATE_g')
end subroutine INTEGRATE_g

subroutine INTEGRATE_a(TIN, TOUT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  real(c_double) :: TIN, TOUT
  integer, parameter :: NVAR = 257, NFIX = 5, NREACT = 979              ! aer_Parameters.h:28-49
  real(c_double), target :: C(NVAR + NFIX)
  real(c_double) :: RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
  common /GDATA_a/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  if (mistra_chem_integrate_common(1_c_int, c_loc(C), TIN, TOUT) /= 0) call mistra_chem_fail('INTEGRATE_a')
end subroutine INTEGRATE_a

subroutine INTEGRATE_t(TIN, TOUT)
  use iso_c_binding
  use mistra_chem_c_api
  implicit none
  real(c_double) :: TIN, TOUT
  integer, parameter :: NVAR = 417, NFIX = 7, NREACT = 1627             ! tot_Parameters.h:28-49
  real(c_double), target :: C(NVAR + NFIX)
  real(c_double) :: RCONST(NREACT), TIME, DT, ATOL(NVAR), RTOL(NVAR), STEPMIN, STEPMAX
  common /GDATA_t/ C, RCONST, TIME, DT, ATOL, RTOL, STEPMIN, STEPMAX
  if (mistra_chem_integrate_common(2_c_int, c_loc(C), TIN, TOUT) /= 0) call mistra_chem_fail('INTEGRATE_t')
end subroutine INTEGRATE_t
