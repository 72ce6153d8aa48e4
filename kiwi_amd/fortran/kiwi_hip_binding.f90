! kiwi_hip_binding.f90 -- iso_c_binding interface to the C-ABI in include/kiwi_hip.h.
!
! This is the Fortran side of the drop-in boundary: a host written in Fortran (the reference's
! minimizer_engine.f90, or this repo's kiwi_amd/fortran/minimizer_hip.f90) `use`s this module and
! replaces its calls to make_seismogram / receiver_scaled_seismograms_to_probes /
! receiver_calculate_misfits by kiwi_hip_* calls (see INTEGRATION.md).  Every function returns
! 0 on success; on failure kiwi_hip_error_message() gives the text for error() (util.f90:133-145).

module kiwi_hip_binding

    use iso_c_binding
    implicit none

    interface

        integer(c_int) function kiwi_hip_init( device, ctx ) bind(C, name='kiwi_hip_init')
            import :: c_int, c_ptr
            integer(c_int), value :: device
            type(c_ptr), intent(out) :: ctx
        end function

        ! one context over ndev_wanted devices of this process (<= 0: all): setters repeated on every device, the trial list of
        ! kiwi_hip_misfits_for_params sharded over them
        integer(c_int) function kiwi_hip_init_multi( ndev_wanted, ctx ) bind(C, name='kiwi_hip_init_multi')
            import :: c_int, c_ptr
            integer(c_int), value :: ndev_wanted
            type(c_ptr), intent(out) :: ctx
        end function

        integer(c_int) function kiwi_hip_ndevices( ctx, n ) bind(C, name='kiwi_hip_ndevices')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), intent(out) :: n
        end function

        ! arithmetic contract of the accumulate kernels: 0 exact (default, bit-identical to the reference's rounding), 1 fused
        integer(c_int) function kiwi_hip_set_arithmetic( ctx, mode ) bind(C, name='kiwi_hip_set_arithmetic')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: mode
        end function

        integer(c_int) function kiwi_hip_get_arithmetic( ctx, mode ) bind(C, name='kiwi_hip_get_arithmetic')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), intent(out) :: mode
        end function

        integer(c_int) function kiwi_hip_destroy( ctx ) bind(C, name='kiwi_hip_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
        end function

        integer(c_int) function kiwi_hip_last_error( ctx, buf, buflen ) bind(C, name='kiwi_hip_last_error')
            import :: c_int, c_ptr, c_char
            type(c_ptr), value :: ctx
            character(kind=c_char), intent(out) :: buf(*)
            integer(c_int), value :: buflen
        end function

        integer(c_int) function kiwi_hip_set_gfdb( ctx, nx, nz, ng, L, dt, dx, dz, firstx, firstz, G, first, nsamp ) &
                bind(C, name='kiwi_hip_set_gfdb')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: nx, nz, ng, L
            real(c_float), value :: dt, dx, dz, firstx, firstz
            real(c_float), intent(in) :: G(*)            ! (L, ng, nz, nx) in Fortran order
            integer(c_int), intent(in) :: first(*), nsamp(*)   ! (ng, nz, nx)
        end function

        integer(c_int) function kiwi_hip_set_interp( ctx, bilinear, xus, zus ) bind(C, name='kiwi_hip_set_interp')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: bilinear, xus, zus
        end function

        integer(c_int) function kiwi_hip_set_effective_dt( ctx, edt ) bind(C, name='kiwi_hip_set_effective_dt')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            real(c_float), value :: edt
        end function

        integer(c_int) function kiwi_hip_set_source_location( ctx, lat, lon, reftime ) &
                bind(C, name='kiwi_hip_set_source_location')
            import :: c_int, c_ptr, c_float, c_double
            type(c_ptr), value :: ctx
            real(c_float), value :: lat, lon              ! degrees, as on the wire (minimizer.f90:511)
            real(c_double), value :: reftime
        end function

        integer(c_int) function kiwi_hip_set_receivers( ctx, nrec, lat, lon, depth, components ) &
                bind(C, name='kiwi_hip_set_receivers')
            import :: c_int, c_ptr, c_float, c_double
            type(c_ptr), value :: ctx
            integer(c_int), value :: nrec
            real(c_double), intent(in) :: lat(*), lon(*)  ! degrees, as in the receivers file
            real(c_float), intent(in) :: depth(*)
            type(c_ptr), intent(in) :: components(*)      ! nrec pointers to NUL-terminated strings
        end function

        integer(c_int) function kiwi_hip_switch_receiver( ctx, irec, enabled ) bind(C, name='kiwi_hip_switch_receiver')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec, enabled
        end function

        integer(c_int) function kiwi_hip_set_reference( ctx, irec, icomp, first, n, data ) &
                bind(C, name='kiwi_hip_set_reference')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec, icomp, first, n
            real(c_float), intent(in) :: data(*)
        end function

        integer(c_int) function kiwi_hip_set_taper( ctx, irec, npts, x, y ) bind(C, name='kiwi_hip_set_taper')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec, npts
            real(c_float), intent(in) :: x(*), y(*)
        end function

        integer(c_int) function kiwi_hip_set_filter( ctx, irec, npts, x, y ) bind(C, name='kiwi_hip_set_filter')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec, npts
            real(c_float), intent(in) :: x(*), y(*)
        end function

        integer(c_int) function kiwi_hip_set_misfit_method( ctx, method ) bind(C, name='kiwi_hip_set_misfit_method')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: method
        end function

        integer(c_int) function kiwi_hip_shift_ref_seismogram( ctx, irec, shift ) bind(C, name='kiwi_hip_shift_ref_seismogram')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec
            real(c_float), value :: shift
        end function

        integer(c_int) function kiwi_hip_autoshift_ref_seismogram( ctx, irec, min_shift, max_shift, isrc, shifts ) &
                bind(C, name='kiwi_hip_autoshift_ref_seismogram')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec, isrc
            real(c_float), value :: min_shift, max_shift
            real(c_float), intent(out) :: shifts(*)
        end function

        integer(c_int) function kiwi_hip_set_floating_shiftrange( ctx, irec, min_shift, max_shift ) &
                bind(C, name='kiwi_hip_set_floating_shiftrange')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec
            real(c_float), value :: min_shift, max_shift
        end function

        integer(c_int) function kiwi_hip_get_floating_shifts( ctx, isrc0, nsrc, shifts ) &
                bind(C, name='kiwi_hip_get_floating_shifts')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc0, nsrc
            real(c_float), intent(out) :: shifts(*)
        end function

        integer(c_int) function kiwi_hip_set_synthetics_factor( ctx, factor ) &
                bind(C, name='kiwi_hip_set_synthetics_factor')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            real(c_float), value :: factor
        end function

        integer(c_int) function kiwi_hip_source_nparams( sourcetype ) bind(C, name='kiwi_hip_source_nparams')
            import :: c_int
            integer(c_int), value :: sourcetype
        end function

        integer(c_int) function kiwi_hip_discretize( sourcetype, params, nparams, effective_dt, cent, maxcent, &
                                                     ncent, moment, risetime ) bind(C, name='kiwi_hip_discretize')
            import :: c_int, c_float
            integer(c_int), value :: sourcetype, nparams, maxcent
            real(c_float), intent(in) :: params(*)
            real(c_float), value :: effective_dt
            real(c_float), intent(out) :: cent(10,*)
            integer(c_int), intent(out) :: ncent
            real(c_float), intent(out) :: moment, risetime
        end function

        ! crust profiles are 31 reals: vp(8) vs(8) rho(8) thickness(7)  (t_crust2x2_1d_profile, crust2x2.f90:45-50)
        integer(c_int) function kiwi_hip_set_source_crust( ctx, rupture_profile, origin_profile ) &
                bind(C, name='kiwi_hip_set_source_crust')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            real(c_float), intent(in) :: rupture_profile(31), origin_profile(31)
        end function

        integer(c_int) function kiwi_hip_set_source_crustal_thickness_limit( ctx, limit ) &
                bind(C, name='kiwi_hip_set_source_crustal_thickness_limit')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            real(c_float), value :: limit
        end function

        integer(c_int) function kiwi_hip_get_source_crustal_thickness( ctx, thickness ) &
                bind(C, name='kiwi_hip_get_source_crustal_thickness')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            real(c_float), intent(out) :: thickness
        end function

        integer(c_int) function kiwi_hip_set_source_constraints( ctx, n, points, normals ) &
                bind(C, name='kiwi_hip_set_source_constraints')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: n
            real(c_float), intent(in) :: points(3,*), normals(3,*)
        end function

        integer(c_int) function kiwi_hip_discretize_eikonal( sourcetype, params, nparams, effective_dt, rupture_profile, &
                                                             ncon, points, normals, cent, maxcent, ncent, moment, &
                                                             risetime ) bind(C, name='kiwi_hip_discretize_eikonal')
            import :: c_int, c_float
            integer(c_int), value :: sourcetype, nparams, ncon, maxcent
            real(c_float), intent(in) :: params(*), rupture_profile(31), points(3,*), normals(3,*)
            real(c_float), value :: effective_dt
            real(c_float), intent(out) :: cent(10,*)
            integer(c_int), intent(out) :: ncent
            real(c_float), intent(out) :: moment, risetime
        end function

        integer(c_int) function kiwi_hip_set_sources( ctx, nsrc, cent_ofs, cent, moment, risetime ) &
                bind(C, name='kiwi_hip_set_sources')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: nsrc
            integer(c_int), intent(in) :: cent_ofs(*)     ! (nsrc+1), 0-based row offsets
            real(c_float), intent(in) :: cent(10,*), moment(*), risetime(*)
        end function

        integer(c_int) function kiwi_hip_set_sources_params( ctx, sourcetype, nsrc, params ) &
                bind(C, name='kiwi_hip_set_sources_params')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: sourcetype, nsrc
            real(c_float), intent(in) :: params(*)        ! (nparams, nsrc)
        end function

        integer(c_int) function kiwi_hip_get_source_status( ctx, isrc0, nsrc, status ) &
                bind(C, name='kiwi_hip_get_source_status')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc0, nsrc          ! isrc0 is 0-based
            integer(c_int), intent(out) :: status(*)      ! 0 ok, 5 empty rupture area, 6 nucleation point outside
        end function

        integer(c_int) function kiwi_hip_source_status_message( code, buf, buflen ) &
                bind(C, name='kiwi_hip_source_status_message')
            import :: c_int, c_char
            integer(c_int), value :: code, buflen
            character(kind=c_char), intent(out) :: buf(*)
        end function

        integer(c_int) function kiwi_hip_misfits_for_params( ctx, sourcetype, nsrc, params, piece, misfit, norm, global, &
                status ) bind(C, name='kiwi_hip_misfits_for_params')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: sourcetype, nsrc, piece   ! piece <= 0: library default
            real(c_float), intent(in) :: params(*)             ! (nparams, nsrc)
            real(c_float), intent(out) :: misfit(*), norm(*)   ! (nmis, nsrc)
            real(c_float), intent(out) :: global(*)            ! (nsrc)
            integer(c_int), intent(out) :: status(*)           ! (nsrc)
        end function

        integer(c_int) function kiwi_hip_effective_cpus() bind(C, name='kiwi_hip_effective_cpus')
            import :: c_int
        end function

        integer(c_int) function kiwi_hip_eikonal_cache_stats( hits, misses, reset ) bind(C, name='kiwi_hip_eikonal_cache_stats')
            import :: c_int, c_long_long
            integer(c_long_long), intent(out) :: hits, misses
            integer(c_int), value :: reset
        end function

      ! eikonal_solver_fmm (eikonal.f90:29) on its own: speed, times are (nx,ny) arrays
        integer(c_int) function kiwi_hip_fast_marching( speed, nx, ny, origin, delta, start, discard, plain, times, &
                                                        fallbacks ) bind(C, name='kiwi_hip_fast_marching')
            import :: c_int, c_float, c_long_long
            real(c_float), intent(in) :: speed(*), origin(2), delta(2), start(2)
            integer(c_int), value :: nx, ny, plain
            real(c_float), value :: discard
            real(c_float), intent(out) :: times(*)
            integer(c_long_long), intent(out) :: fallbacks
        end function

        integer(c_int) function kiwi_hip_eval( ctx, isrc0, nsrc ) bind(C, name='kiwi_hip_eval')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc0, nsrc          ! isrc0 is 0-based
        end function

        integer(c_int) function kiwi_hip_sync( ctx ) bind(C, name='kiwi_hip_sync')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
        end function

        integer(c_int) function kiwi_hip_nmisfits( ctx, nmis ) bind(C, name='kiwi_hip_nmisfits')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), intent(out) :: nmis
        end function

        ! device pointer of the global misfits of a source range (for a device-to-device all-gather)
        integer(c_int) function kiwi_hip_get_global_misfits_device( ctx, isrc0, nsrc, device_ptr ) bind(C, name='kiwi_hip_get_global_misfits_device')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc0, nsrc
            type(c_ptr), intent(out) :: device_ptr
        end function

        integer(c_int) function kiwi_hip_get_misfits( ctx, isrc0, nsrc, misfit, norm, global ) &
                bind(C, name='kiwi_hip_get_misfits')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc0, nsrc
            real(c_float), intent(out) :: misfit(*), norm(*), global(*)   ! (nmis,nsrc), (nmis,nsrc), (nsrc)
        end function

        integer(c_int) function kiwi_hip_set_keep_synthetics( ctx, which ) bind(C, name='kiwi_hip_set_keep_synthetics')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: which
        end function

        integer(c_int) function kiwi_hip_get_synthetics( ctx, isrc, irec, icomp, which, first, n, out, maxn ) &
                bind(C, name='kiwi_hip_get_synthetics')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc, irec, icomp, which, maxn
            integer(c_int), intent(out) :: first, n
            real(c_float), intent(out) :: out(*)
        end function

        integer(c_int) function kiwi_hip_get_kernel_ms( ctx, ms, launches ) bind(C, name='kiwi_hip_get_kernel_ms')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            real(c_float), intent(out) :: ms(4)
            integer(c_int), intent(out) :: launches(3)
        end function

        integer(c_int) function kiwi_hip_get_geometry( ctx, isrc, irec, maxcent, ncent, records ) &
                bind(C, name='kiwi_hip_get_geometry')
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc, irec, maxcent
            integer(c_int), intent(out) :: ncent
            type(c_ptr), value :: records                    ! maxcent records of 80 bytes (kiwi_kernels.hpp, GeoRec)
        end function

      ! fcn: a bind(C) function  integer(c_int) f( user, k, m, n, xs, fv )  with value arguments user (c_ptr), k, m, n and
      ! arrays xs(n,k), fv(m,k); pass c_funloc(f)
        integer(c_int) function kiwi_hip_lmdif( fcn, user, m, n, x, fvec, ftol, xtol, gtol, maxfev, epsfcn, diag, mode, factor, &
                                                info, nfev ) bind(C, name='kiwi_hip_lmdif')
            import :: c_int, c_ptr, c_funptr, c_float
            type(c_funptr), value :: fcn
            type(c_ptr), value :: user
            integer(c_int), value :: m, n, maxfev, mode
            real(c_float), value :: ftol, xtol, gtol, epsfcn, factor
            real(c_float), intent(inout) :: x(*), fvec(*), diag(*)
            integer(c_int), intent(out) :: info, nfev
        end function

        integer(c_int) function kiwi_hip_get_amp_spectrum( ctx, isrc, irec, icomp, which_probe, filtered, df, n, out, maxn ) &
                bind(C, name='kiwi_hip_get_amp_spectrum')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc, irec, icomp, which_probe, filtered, maxn
            real(c_float), intent(out) :: df
            integer(c_int), intent(out) :: n
            real(c_float), intent(out) :: out(*)
        end function

        integer(c_int) function kiwi_hip_principal_axes( sourcetype, params, pax, tax ) bind(C, name='kiwi_hip_principal_axes')
            import :: c_int, c_float
            integer(c_int), value :: sourcetype
            real(c_float), intent(in) :: params(*)
            real(c_float), intent(out) :: pax(2), tax(2)
        end function

        integer(c_int) function kiwi_hip_get_cross_correlations( ctx, isrc, irec, min_shift, max_shift, first_shift, nshift, &
                                                                  cc, maxn ) bind(C, name='kiwi_hip_get_cross_correlations')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc, irec, maxn
            real(c_float), value :: min_shift, max_shift
            integer(c_int), intent(out) :: first_shift, nshift
            real(c_float), intent(out) :: cc(*)
        end function

        integer(c_int) function kiwi_hip_get_peak_amplitudes( ctx, isrc, differentiate, out ) &
                bind(C, name='kiwi_hip_get_peak_amplitudes')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc, differentiate
            real(c_float), intent(out) :: out(*)
        end function

        integer(c_int) function kiwi_hip_get_arias_intensities( ctx, isrc, out ) bind(C, name='kiwi_hip_get_arias_intensities')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc
            real(c_float), intent(out) :: out(*)
        end function

        integer(c_int) function kiwi_hip_minimize_lm( ctx, sourcetype, params, mask, mins, maxs, info, iterations, misfit, best ) &
                bind(C, name='kiwi_hip_minimize_lm')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: sourcetype
            real(c_float), intent(inout) :: params(*)
            integer(c_int), intent(in) :: mask(*)
            type(c_ptr), value :: mins, maxs                 ! real(c_float) arrays or c_null_ptr
            integer(c_int), intent(out) :: info, iterations
            real(c_float), intent(out) :: misfit
            real(c_float), intent(out) :: best(*)
        end function

        integer(c_int) function kiwi_hip_get_source_centroids( ctx, isrc, maxcent, ncent, cent ) &
                bind(C, name='kiwi_hip_get_source_centroids')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: isrc, maxcent
            integer(c_int), intent(out) :: ncent
            real(c_float), intent(out) :: cent(10,*)
        end function

        integer(c_int) function kiwi_hip_get_device_bytes( ctx, bytes ) bind(C, name='kiwi_hip_get_device_bytes')
            import :: c_int, c_ptr, c_long_long
            type(c_ptr), value :: ctx
            integer(c_long_long), intent(out) :: bytes
        end function

        integer(c_int) function kiwi_hip_build_flags( buf, buflen ) bind(C, name='kiwi_hip_build_flags')
            import :: c_int, c_char
            character(kind=c_char), intent(out) :: buf(*)
            integer(c_int), value :: buflen
        end function

      ! diagnostics: GB/s of a pure read of `bytes` bytes of device memory
        integer(c_int) function kiwi_hip_measure_read_bandwidth( ctx, bytes, reps, gbs ) &
                bind(C, name='kiwi_hip_measure_read_bandwidth')
            import :: c_int, c_ptr, c_long_long, c_double
            type(c_ptr), value :: ctx
            integer(c_long_long), value :: bytes
            integer(c_int), value :: reps
            real(c_double), intent(out) :: gbs
        end function

        integer(c_int) function kiwi_hip_get_reference( ctx, irec, icomp, which, first, n, out, maxn ) &
                bind(C, name='kiwi_hip_get_reference')
            import :: c_int, c_ptr, c_float
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec, icomp, which, maxn
            integer(c_int), intent(out) :: first, n
            real(c_float), intent(out) :: out(*)
        end function

        integer(c_int) function kiwi_hip_get_receiver_geometry( ctx, irec, azi, bazi, dist ) &
                bind(C, name='kiwi_hip_get_receiver_geometry')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: ctx
            integer(c_int), value :: irec
            real(c_double), intent(out) :: azi, bazi, dist
        end function

    end interface

  contains

  ! the library's last error as a Fortran string, for error() / "<cmd>: nok >" answers
    function kiwi_hip_error_message( ctx ) result( msg )
        type(c_ptr), intent(in) :: ctx
        character(len=:), allocatable :: msg
        character(kind=c_char) :: buf(1024)
        integer :: i, n, rc
        rc = kiwi_hip_last_error( ctx, buf, 1024_c_int )
        n = 0
        do i = 1, 1024
            if (buf(i) == c_null_char) exit
            n = i
        end do
        allocate( character(len=n) :: msg )
        do i = 1, n
            msg(i:i) = buf(i)
        end do
    end function

end module
