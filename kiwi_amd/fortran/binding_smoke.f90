! binding_smoke.f90 -- proves that a Fortran host reaches the C-ABI through iso_c_binding:
! discretises the reference's own unit-test source (test_source_bilat.f90:47-60) through
! kiwi_hip_discretize (host-side, needs no GPU) and prints the result for tests/test_fortran_binding.py.
! With a GPU present it also opens and closes a device context.
program binding_smoke

    use iso_c_binding
    use kiwi_hip_binding
    implicit none

    real(c_float) :: params(14), cent(10,4096), moment, risetime, msum
    integer(c_int) :: ncent, rc
    type(c_ptr) :: ctx
    integer :: i

    params = (/ 0., 0., 0., 1000., 1., 90., 45., 90., 0., 2000., 0., 1000., 2000., 1. /)
    if (kiwi_hip_source_nparams( 1_c_int ) /= 14) stop 2
    rc = kiwi_hip_discretize( 1_c_int, params, 14_c_int, 0.5_c_float, cent, 4096_c_int, ncent, moment, risetime )
    if (rc /= 0) stop 3
    msum = 0.
    do i = 1, ncent
        msum = msum + cent(5,i)
    end do
    print '(a,i6,a,f10.6,a,f8.3)', 'ncent ', ncent, ' sum_mxx ', msum, ' moment ', moment

    rc = kiwi_hip_init( 0_c_int, ctx )
    if (rc == 0) then
        print '(a)', 'device context ok'
        rc = kiwi_hip_destroy( ctx )
    else
        print '(a,a)', 'no device: ', kiwi_hip_error_message( c_null_ptr )
    end if

end program
