! ref_driver.f90 -- TEST INFRASTRUCTURE (not product code).
!
! bind(C) entry points that do nothing but CALL the reference's own, unmodified
! Fortran modules (compiled from /root/reference by oracle/Makefile target 'ref')
! so that tests can compare oracle/libko.so against the real thing bit for bit.
! No reference source is copied here; every routine below is a thin argument
! adapter around one or two reference procedures, named in its comment.

module ref_driver

    use iso_c_binding
    use constants
    use util
    use sparse_trace
    use piecewise_linear_function
    use orthodrome
    use euler
    use discrete_source
    use parameterized_source
    use source_moment_tensor
    use source_bilat
    use source_circular
    use source_point_lp
    use source_eikonal
    use source_mt_eikonal
    use crust2x2
    use geometry
    use eikonal
    use better_varying_string

    implicit none

    abstract interface
        function lm_cfcn_t( m, n, x, fvec ) bind(C) result(iflag)
            import :: c_int, c_float
            integer(c_int), value :: m, n
            real(c_float) :: x(*), fvec(*)
            integer(c_int) :: iflag
        end function
    end interface
    procedure(lm_cfcn_t), pointer, save :: lm_cfcn => null()

  contains

  ! sminpack/lmdif.f (the reference's single-precision MINPACK, compiled as it is) with a C residual callback
    subroutine ref_lmdif( cfcn, m, n, x, fvec, ftol, xtol, gtol, maxfev, epsfcn, diag, mode, factor, info, nfev ) &
                          bind(C, name='ref_lmdif')
        type(c_funptr), value :: cfcn
        integer(c_int), value :: m, n, maxfev, mode
        real(c_float), value :: ftol, xtol, gtol, epsfcn, factor
        real(c_float) :: x(n), fvec(m), diag(n)
        integer(c_int), intent(out) :: info, nfev
        real :: fjac(m,n), qtf(n), wa1(n), wa2(n), wa3(n), wa4(m)
        integer :: ipvt(n)
        external lmdif
        call c_f_procpointer( cfcn, lm_cfcn )
        call lmdif( lm_trampoline, m, n, x, fvec, ftol, xtol, gtol, maxfev, epsfcn, diag, mode, factor, 0, info, nfev, &
                    fjac, m, ipvt, qtf, wa1, wa2, wa3, wa4 )
    end subroutine

    subroutine lm_trampoline( m, n, x, fvec, iflag )
        integer :: m, n, iflag
        real :: x(n), fvec(m)
        integer :: rc
        rc = lm_cfcn( m, n, x, fvec )
        if (rc < 0) iflag = rc
    end subroutine

  ! trace_pack (sparse_trace.f90:443): returns number of strips and their spans
    subroutine ref_trace_pack( lo, n, data, maxstrips, nstrips, spans, tspan ) bind(C, name='ref_trace_pack')
        integer(c_int), value :: lo, n, maxstrips
        real(c_float), intent(in) :: data(n)
        integer(c_int), intent(out) :: nstrips, spans(2,maxstrips), tspan(2)
        type(t_strip) :: s
        type(t_trace) :: t
        integer :: i
        call strip_init( (/lo,lo+n-1/), data, s )
        call trace_pack( s, t )
        nstrips = t%nstrips
        tspan = t%span
        do i=1,min(nstrips,maxstrips)
            spans(:,i) = strip_span(t%strips(i))
        end do
        call trace_destroy(t)
        call strip_destroy(s)
    end subroutine

  ! trace_pack + trace_multiply_add (sparse_trace.f90:597) onto an optional existing strip.
  ! mode: 0 none, 1 itraceshift_, 2 rtraceshift_
    subroutine ref_multiply_add( tlo, tn, tdata, has_s, slo, sn, sdata, factor, mode, ishift, rshift, &
                                 omax, olo, on, odata ) bind(C, name='ref_multiply_add')
        integer(c_int), value :: tlo, tn, has_s, slo, sn, mode, ishift, omax
        real(c_float), value :: factor, rshift
        real(c_float), intent(in) :: tdata(tn), sdata(*)
        integer(c_int), intent(out) :: olo, on
        real(c_float), intent(out) :: odata(omax)
        type(t_strip) :: ts, s
        type(t_trace) :: t
        call strip_init( (/tlo,tlo+tn-1/), tdata, ts )
        call trace_pack( ts, t )
        if (has_s /= 0) call strip_init( (/slo,slo+sn-1/), sdata(1:sn), s )
        if (mode == 0) then
            call trace_multiply_add( t, s, factor )
        else if (mode == 1) then
            call trace_multiply_add( t, s, factor, itraceshift_=ishift )
        else
            call trace_multiply_add( t, s, factor, rtraceshift_=rshift )
        end if
        olo = lbound(s%data,1)
        on = size(s%data)
        odata(1:min(on,omax)) = s%data(olo:olo+min(on,omax)-1)
        call trace_destroy(t)
        call strip_destroy(ts)
        call strip_destroy(s)
    end subroutine

  ! the summation of gfdb_get_trace_bilin (gfdb.f90:944-949) on its real primitive
  ! trace_multiply_add_nogrow (sparse_trace.f90:710): four packed traces blended
  ! over the union of their spans with the weights formed exactly as gfdb.f90 forms them.
    subroutine ref_blend4( lo, n, nmax, data, dix, diz, olo, on, odata ) bind(C, name='ref_blend4')
        integer(c_int), intent(in) :: lo(4), n(4)
        integer(c_int), value :: nmax
        real(c_float), intent(in) :: data(nmax,4)
        real(c_float), value :: dix, diz
        integer(c_int), intent(out) :: olo, on
        real(c_float), intent(out) :: odata(*)
        type(t_strip) :: s
        type(t_trace) :: t(4)
        integer :: i
        integer, dimension(2) :: span
        real, dimension(:), allocatable :: buf
        do i=1,4
            call strip_init( (/lo(i),lo(i)+n(i)-1/), data(1:n(i),i), s )
            call trace_pack( s, t(i) )
        end do
        span(1) = min( t(1)%span(1), t(2)%span(1), t(3)%span(1), t(4)%span(1) )
        span(2) = max( t(1)%span(2), t(2)%span(2), t(3)%span(2), t(4)%span(2) )
        allocate( buf(span(1):span(2)) )
        buf(:) = 0.
        call trace_multiply_add_nogrow( t(1), buf, span, (1.-dix)*(1.-diz) )
        call trace_multiply_add_nogrow( t(2), buf, span, (1.-dix)*diz )
        call trace_multiply_add_nogrow( t(3), buf, span, dix*(1.-diz) )
        call trace_multiply_add_nogrow( t(4), buf, span, dix*diz )
        olo = span(1)
        on = span(2)-span(1)+1
        odata(1:on) = buf(:)
        deallocate(buf)
        do i=1,4
            call trace_destroy(t(i))
        end do
        call strip_destroy(s)
    end subroutine

  ! strip_dataspan (sparse_trace.f90:347)
    subroutine ref_strip_dataspan( lo, n, data, ds ) bind(C, name='ref_strip_dataspan')
        integer(c_int), value :: lo, n
        real(c_float), intent(in) :: data(n)
        integer(c_int), intent(out) :: ds(2)
        type(t_strip) :: s
        call strip_init( (/lo,lo+n-1/), data, s )
        ds = strip_dataspan(s)
        call strip_destroy(s)
    end subroutine

  ! strip_fold (sparse_trace.f90:379)
    subroutine ref_strip_fold( lo, n, data, nshifts, shifts, amps, omax, olo, on, odata ) bind(C, name='ref_strip_fold')
        integer(c_int), value :: lo, n, nshifts, omax
        real(c_float), intent(in) :: data(n), shifts(nshifts), amps(nshifts)
        integer(c_int), intent(out) :: olo, on
        real(c_float), intent(out) :: odata(omax)
        type(t_strip) :: s
        call strip_init( (/lo,lo+n-1/), data, s )
        call strip_fold( s, shifts, amps )
        olo = lbound(s%data,1)
        on = size(s%data)
        odata(1:min(on,omax)) = s%data(olo:olo+min(on,omax)-1)
        call strip_destroy(s)
    end subroutine

  ! d2r (orthodrome.f90:313-338)
    function ref_d2r_d( deg ) bind(C, name='ref_d2r_d') result(rad)
        real(c_double), value :: deg
        real(c_double) :: rad
        rad = d2r(deg)
    end function
    function ref_d2r_r( deg ) bind(C, name='ref_d2r_r') result(rad)
        real(c_float), value :: deg
        real(c_float) :: rad
        rad = d2r(deg)
    end function

  ! azibazi + distance_accurate50m (orthodrome.f90:245,193); radians in
    subroutine ref_azibazi_dist( alat, alon, blat, blon, azi, bazi, dist ) bind(C, name='ref_azibazi_dist')
        real(c_double), value :: alat, alon, blat, blon
        real(c_double), intent(out) :: azi, bazi, dist
        type(t_geo_coords) :: a, b
        a%lat = alat; a%lon = alon; b%lat = blat; b%lon = blon
        call azibazi( a, b, azi, bazi )
        dist = distance_accurate50m( a, b )
    end subroutine

  ! approx_differential_azidist (orthodrome.f90:77)
    subroutine ref_approx_differential_azidist( dx, dy, azi, bazi, dist, nazi, nbazi, ndist ) &
                                 bind(C, name='ref_approx_differential_azidist')
        real(c_float), value :: dx, dy
        real(c_double), value :: azi, bazi, dist
        real(c_double), intent(out) :: nazi, nbazi, ndist
        call approx_differential_azidist( dx, dy, azi, bazi, dist, nazi, nbazi, ndist )
    end subroutine

  ! init_euler (euler.f90:28); mat returned column-major (Fortran order)
    subroutine ref_init_euler( alpha, beta, gamma, mat ) bind(C, name='ref_init_euler')
        real(c_float), value :: alpha, beta, gamma
        real(c_float), intent(out) :: mat(3,3)
        call init_euler( alpha, beta, gamma, mat )
    end subroutine

  ! plf_integrate_and_centroid (piecewise_linear_function.f90:165)
    subroutine ref_plf_integrate_and_centroid( n, x, y, a, b, area, centroid ) bind(C, name='ref_plf_integrate_and_centroid')
        integer(c_int), value :: n
        real(c_float), intent(in) :: x(n), y(n)
        real(c_float), value :: a, b
        real(c_float), intent(out) :: area, centroid
        type(t_plf) :: s
        call plf_make( s, x, y )
        call plf_integrate_and_centroid( s, a, b, area, centroid )
        call plf_destroy( s )
    end subroutine

  ! plf_taper_array real variant (piecewise_linear_function.f90:195); ip 0 cos, 1 linear, 2 zero_one
    subroutine ref_plf_taper_array_r( n, x, y, lo, hi, array, dx, ip ) bind(C, name='ref_plf_taper_array_r')
        integer(c_int), value :: n, lo, hi, ip
        real(c_float), intent(in) :: x(n), y(n)
        real(c_float), intent(inout) :: array(lo:hi)
        real(c_float), value :: dx
        type(t_plf) :: s
        call plf_make( s, x, y )
        if (ip == 0) then
            call plf_taper_array( s, array, (/lo,hi/), dx, ip_cos )
        else if (ip == 1) then
            call plf_taper_array( s, array, (/lo,hi/), dx, ip_linear )
        else
            call plf_taper_array( s, array, (/lo,hi/), dx, ip_zero_one )
        end if
        call plf_destroy( s )
    end subroutine

  ! psm_set_<type> + psm_to_tdsm_<type> (source_bilat.f90:173,241; source_circular.f90:165,235;
  ! source_moment_tensor.f90:163,205).  sourcetype ids as parameterized_source.f90:45-50.
  ! psm_set_bilat -> psm%pax, psm%tax (source_bilat.f90:216-239): P and T axis of a bilateral source
    subroutine ref_principal_axes_bilat( params, pax, tax ) bind(C, name='ref_principal_axes_bilat')
        real(c_float), intent(in) :: params(14)
        real(c_float), intent(out) :: pax(2), tax(2)
        type(t_psm), save :: psm
        logical :: omc
        call psm_destroy( psm )
        call psm_set_bilat( psm, params, .false., omc )
        pax = psm%pax
        tax = psm%tax
    end subroutine

    subroutine ref_discretize( sourcetype, np, params, effective_dt, maxc, nc, cent, moment, risetime, &
                               grid_size ) bind(C, name='ref_discretize')
        integer(c_int), value :: sourcetype, np, maxc
        real(c_float), intent(in) :: params(np)
        real(c_float), value :: effective_dt
        integer(c_int), intent(out) :: nc, grid_size(3)
        real(c_float), intent(out) :: cent(10,maxc), moment, risetime
        type(t_psm), save :: psm
        type(t_tdsm) :: tdsm
        logical :: omc, ok
        integer :: i
        call psm_destroy( psm )
        ok = .false.
        grid_size = 0
        if (sourcetype == psm_bilat) then
            call psm_set_bilat( psm, params, .false., omc )
            psm%sourcetype = psm_bilat
            call psm_to_tdsm_bilat( psm, tdsm, effective_dt, ok )
        else if (sourcetype == psm_circular) then
            call psm_set_circular( psm, params, .false., omc )
            psm%sourcetype = psm_circular
            call psm_to_tdsm_circular( psm, tdsm, effective_dt, ok )
        else if (sourcetype == psm_moment_tensor) then
            call psm_set_moment_tensor( psm, params, .false., omc )
            psm%sourcetype = psm_moment_tensor
            call psm_to_tdsm_moment_tensor( psm, tdsm, effective_dt, ok )
        else if (sourcetype == psm_point_lp) then
            call psm_set_point_lp( psm, params, .false., omc )
            psm%sourcetype = psm_point_lp
            call psm_to_tdsm_point_lp( psm, tdsm, effective_dt, ok )
        end if
        nc = -1
        if (.not. ok) return
        nc = size(tdsm%centroids)
        moment = psm%moment
        risetime = psm%risetime
        grid_size(1:size(psm%grid_size)) = psm%grid_size
        do i=1,min(nc,maxc)
            cent(1,i) = tdsm%centroids(i)%north
            cent(2,i) = tdsm%centroids(i)%east
            cent(3,i) = tdsm%centroids(i)%depth
            cent(4,i) = tdsm%centroids(i)%time
            cent(5:10,i) = tdsm%centroids(i)%m(:)
        end do
        call tdsm_destroy( tdsm )
    end subroutine


  ! crust2x2_load + crust2x2_get_profile (crust2x2.f90:76-105): 1-D profile at the location GIVEN AS IS
  ! (the reference calls it both with degrees and, in psm_make_*_grid, with radians)
    subroutine ref_crust_profile( dir, ndir, lat, lon, vp, vs, rho, thickness, ok ) bind(C, name='ref_crust_profile')
        integer(c_int), value :: ndir
        character(kind=c_char), intent(in) :: dir(ndir)
        real(c_double), value :: lat, lon
        real(c_float), intent(out) :: vp(8), vs(8), rho(8), thickness(7)
        integer(c_int), intent(out) :: ok
        type(t_crust2x2_1d_profile) :: profile
        type(t_geo_coords) :: loc
        character(len=ndir) :: d
        logical :: lok
        integer :: i
        ok = 1
        if (.not. crust2x2_loaded) then
            do i=1,ndir
                d(i:i) = dir(i)
            end do
            call crust2x2_load( d, lok )
            if (.not. lok) then
                ok = 0
                return
            end if
        end if
        loc%lat = lat; loc%lon = lon
        call crust2x2_get_profile( loc, profile )
        vp = profile%vp; vs = profile%vs; rho = profile%rho; thickness = profile%thickness
    end subroutine

  ! eikonal_solver_fmm (eikonal.f90:29)
    subroutine ref_eikonal_fmm( nx, ny, speed, origin, delta, initialpoint, times ) bind(C, name='ref_eikonal_fmm')
        integer(c_int), value :: nx, ny
        real(c_float), intent(in) :: speed(nx,ny), origin(2), delta(2), initialpoint(2)
        real(c_float), intent(out) :: times(nx,ny)
        call eikonal_solver_fmm( speed, origin, delta, initialpoint, times )
    end subroutine

  ! psm_set_origin_and_time (default constraints from CRUST2.0, parameterized_source.f90:127-145,185-196) +
  ! psm_set_[mt_]eikonal + psm_to_tdsm_[mt_]eikonal (source_eikonal.f90:205,259; source_mt_eikonal.f90:200,266).
  ! lat/lon in radians as the engine stores them; crust2x2 must have been loaded (ref_crust_profile).
    subroutine ref_discretize_eikonal( sourcetype, np, params, effective_dt, lat, lon, thickness_limit, maxc, nc, cent, &
                                       moment, risetime, grid_size, con_points, con_normals ) &
                                       bind(C, name='ref_discretize_eikonal')
        integer(c_int), value :: sourcetype, np, maxc
        real(c_float), intent(in) :: params(np)
        real(c_float), value :: effective_dt, thickness_limit
        real(c_double), value :: lat, lon
        integer(c_int), intent(out) :: nc, grid_size(2)
        real(c_float), intent(out) :: cent(10,maxc), moment, risetime, con_points(3,2), con_normals(3,2)
        type(t_psm), save :: psm
        type(t_tdsm) :: tdsm
        type(t_geo_coords) :: origin
        logical :: omc, ok
        integer :: i
        call psm_destroy( psm )
        origin%lat = lat; origin%lon = lon
        psm%crustal_thickness_limit = thickness_limit
        call psm_set_origin_and_time( psm, origin, 0.d0 )
        do i=1,2
            con_points(:,i) = psm%constraints(i)%point
            con_normals(:,i) = psm%constraints(i)%normal
        end do
        ok = .false.
        if (sourcetype == psm_eikonal) then
            call psm_set_eikonal( psm, params, .false., omc )
            psm%sourcetype = psm_eikonal
            call psm_to_tdsm_eikonal( psm, tdsm, effective_dt, ok )
        else
            call psm_set_mt_eikonal( psm, params, .false., omc )
            psm%sourcetype = psm_mt_eikonal
            call psm_to_tdsm_mt_eikonal( psm, tdsm, effective_dt, ok )
        end if
        nc = -1
        if (.not. ok) return
        nc = size(tdsm%centroids)
        moment = psm%moment
        risetime = psm%risetime
        grid_size(1:2) = psm%grid_size(1:2)
        do i=1,min(nc,maxc)
            cent(1,i) = tdsm%centroids(i)%north
            cent(2,i) = tdsm%centroids(i)%east
            cent(3,i) = tdsm%centroids(i)%depth
            cent(4,i) = tdsm%centroids(i)%time
            cent(5:10,i) = tdsm%centroids(i)%m(:)
        end do
        call tdsm_destroy( tdsm )
    end subroutine

end module
