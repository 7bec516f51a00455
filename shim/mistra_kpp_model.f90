! The ONE shim file that is compiled against the model's own modules and COMMON declarations (put it behind common_modules.f90 and
! mistra_kpp_drive.f90 in src/Makefile): KPP_DRIVE_RUN is what the patched kpp_driver calls behind its layer loop (shim/kpp_drive.patch).
! It names the model arrays the per-layer drivers read and write — module gas_common (common_modules.f90:77-133: s1, s3 and the species
! index maps mk_interface builds, utils.f90:82-140), COMMON /blck17/ (gas.f:101), /budg/ (gas.f:104), /budgs/ (bud_s_g.f:63) — and hands
! them, as they stand in memory, to kpp_drive_run_arrays (shim/mistra_kpp_drive.f90), i.e. to ONE mistra_chem_drive call per mechanism.
subroutine KPP_DRIVE_RUN(tkpp, dt_ch)
  USE gas_common, ONLY : j1, j5, s1, s3, gas_m2k_g, gas_k2m_g, rad_m2k_g, rad_k2m_g, gas_m2k_a, gas_k2m_a, rad_m2k_a, rad_k2m_a, &
       gas_m2k_t, gas_k2m_t, rad_m2k_t, rad_k2m_t
  USE global_params, ONLY : j2, j6, n, nkc, nlev, nrxn
  USE mistra_kpp_drive, ONLY : kpp_drive_maps, kpp_drive_run_arrays
  implicit none
  double precision, intent(in) :: tkpp, dt_ch
  common /blck17/ sl1(j2,nkc,n), sion1(j6,nkc,n)
  double precision :: sl1, sion1
  common /budg/ bg(2,nrxn,nlev), il(nlev)
  double precision :: bg
  integer :: il
  common /budgs/ bgs(2,122,n)
  double precision :: bgs
  logical, save :: first = .true.
  if (first) then
     call kpp_drive_maps(1, j1, gas_m2k_g, gas_k2m_g, j5, rad_m2k_g, rad_k2m_g)
     call kpp_drive_maps(2, j1, gas_m2k_a, gas_k2m_a, j5, rad_m2k_a, rad_k2m_a)
     call kpp_drive_maps(3, j1, gas_m2k_t, gas_k2m_t, j5, rad_m2k_t, rad_k2m_t)
     first = .false.
  end if
  call kpp_drive_run_arrays(tkpp, dt_ch, n, s1, s3, sl1, sion1, nrxn, nlev, il, bg, bgs)
end subroutine KPP_DRIVE_RUN
