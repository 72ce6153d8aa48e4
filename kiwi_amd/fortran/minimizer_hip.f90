! minimizer_hip.f90 -- a Fortran host speaking Kiwi's `minimizer` stdin/stdout command protocol
! (minimizer.f90:1676-1701,1729-1811) on top of the MI355X engine (include/kiwi_hip.h, through
! kiwi_hip_binding.f90).  Written from the protocol description, not from the reference's host:
! one command per line, '#' starts a comment, blanks collapse; answers are "<cmd>: ok",
! "<cmd>: ok >" + answer line, "<cmd>: nok" or "<cmd>: nok >" + error line, flushed after each.
!
! Commands (SURVEY.md 8b's subset for the hot path, widened to most of minimizer.f90:1729-1811):
!   set_database <base>                    <base>.kiwiflat (flat dense GFDB) or the reference's HDF5 <base>.index + chunks
!   set_effective_dt <dt>
!   set_local_interpolation nearest_neighbor|bilinear
!   set_spacial_undersampling <nx> <nz>
!   set_receivers <file> [has_depth]       lines: lat lon [depth] components
!   switch_receiver <i> on|off
!   set_source_location <lat> <lon> <reftime>
!   set_source_crust, set_source_constraints, set_source_crustal_thickness_limit, get_source_crustal_thickness
!   set_source_params <type> p1 .. pn      bilateral | circular | point_lp | eikonal | mt_eikonal | moment_tensor
!   set_source_params_mask T|F ..;  set_source_subparams v ..;  set_source_subparams_limits mins .. maxs ..;
!   get_source_subparams;  minimize_lm     answers "info iterations misfit"; Jacobians are batched on the device
!   set_ref_seismograms <base> table       files <base>-<irec>-<comp>.table (time value)
!   shift_ref_seismogram, autoshift_ref_seismogram
!   set_misfit_method <name>;  set_misfit_taper <i> x y ..;  set_misfit_filter x y ..;  set_misfit_filter_1 <i> x y ..
!   set_floating_shiftrange <i> lo hi;  get_floating_shifts
!   set_synthetics_factor <f>
!   get_misfits;  get_global_misfit
!   output_seismograms <base> table synthetics|references [plain|tapered|filtered]
!   output_distances <file>;  output_source_model <base>
!   get_cached_traces_memory;  set_cached_traces_memory_limit, set_verbose, set_ignore_sigint   accepted, no effect
!   get_peak_amplitudes 1|2;  get_arias_intensities;  output_cross_correlations <base> <shift-min> <shift-max>
!   get_principal_axes                     bilateral sources
!   output_seismogram_spectra <base> synthetics|references plain|filtered
! Not provided: (diagnostics beside the
!   inversion loop) and mseed / sac file formats.
! Batch extension (SURVEY.md 8f-1), one pipe round trip for a whole grid:
!   eval_sources <type> <paramfile> <outfile>   one parameter vector per line in; per source
!                                               "global m1 n1 m2 n2 .." out; answers the number of sources, followed by
!                                               "failed i1 i2 .." when the discretiser rejected some (their rows: zeros)

program minimizer_hip

    use iso_c_binding
    use kiwi_hip_binding
    implicit none

    integer, parameter :: maxline = 262144
    character(len=maxline) :: line, args
    character(len=64) :: command
    character(len=:), allocatable :: answer, errstr
    type(c_ptr) :: ctx
    integer :: iostat, rc
    logical :: ok, have_ctx

  ! engine-side bookkeeping the host needs for file names and answers
    integer :: nreceivers = 0
    character(len=8), allocatable :: components(:)
    logical, allocatable :: enabled(:)
    real(c_float) :: db_dt = 0.
    character(len=16) :: db_format = ''
    real(c_double) :: ref_time = 0.d0
    real(c_float) :: effective_dt = 1.
    logical :: source_set = .false., evaluated = .false.
    integer(c_int) :: cur_bilinear = 0, cur_xus = 1, cur_zus = 1
  ! the current source as psm holds it: type, parameters, mask and limits of the masked ones (minimize_lm)
    integer(c_int) :: cur_st = 0
    real(c_float), allocatable :: cur_params(:)
    real(c_float), allocatable, target :: sub_mins(:), sub_maxs(:)
    integer(c_int), allocatable :: cur_mask(:)

    ctx = c_null_ptr
    have_ctx = .false.

    stdinloop: do
        read (*,'(a)',iostat=iostat) line
        if (iostat /= 0) exit stdinloop
        call reduce_whitespace( line )
        if (len_trim(line) == 0) cycle
        call split_first( line, command, args )
        answer = ''
        errstr = ''
        call do_command( trim(command), trim(args), ok )
        if (ok) then
            if (len(answer) == 0) then
                write (*,'(a)') trim(command)//': ok'
            else
                write (*,'(a)') trim(command)//': ok >'
                write (*,'(a)') answer
            end if
        else
            if (len(errstr) == 0) then
                write (*,'(a)') trim(command)//': nok'
            else
                write (*,'(a)') trim(command)//': nok >'
                write (*,'(a)') errstr
            end if
        end if
        flush (6)
    end do stdinloop

    if (have_ctx) rc = kiwi_hip_destroy( ctx )

  contains

    subroutine reduce_whitespace( s )
        character(len=*), intent(inout) :: s
        character(len=len(s)) :: b
        integer :: i, j
        logical :: ws
        b = ''
        j = 1
        ws = .true.
        do i = 1, len_trim(s)
            if (s(i:i) == '#') exit
            if (s(i:i) /= ' ' .and. s(i:i) /= char(9)) then
                b(j:j) = s(i:i); j = j + 1; ws = .false.
            else if (.not. ws) then
                b(j:j) = ' '; j = j + 1; ws = .true.
            end if
        end do
        s = b
    end subroutine

    subroutine split_first( s, first, rest )
        character(len=*), intent(in) :: s
        character(len=*), intent(out) :: first, rest
        integer :: i
        i = index( trim(s), ' ' )
        if (i == 0) then
            first = trim(s); rest = ''
        else
            first = s(1:i-1); rest = adjustl(s(i+1:))
        end if
    end subroutine

    integer function count_words( s )
        character(len=*), intent(in) :: s
        integer :: i
        logical :: inword
        count_words = 0
        inword = .false.
        do i = 1, len_trim(s)
            if (s(i:i) /= ' ') then
                if (.not. inword) count_words = count_words + 1
                inword = .true.
            else
                inword = .false.
            end if
        end do
    end function

    subroutine fail( msg )
        character(len=*), intent(in) :: msg
        errstr = msg
    end subroutine

  ! turn a C-ABI return code into ok / error text
    logical function check( rc_ )
        integer(c_int), intent(in) :: rc_
        check = (rc_ == 0)
        if (.not. check) errstr = kiwi_hip_error_message( ctx )
    end function

    logical function need_ctx()
        integer(c_int) :: rc_
        need_ctx = .true.
        if (have_ctx) return
        ! KIWI_HIP_NDEV=n: one engine over n devices of this node (0: all of them); the trial list of eval_sources is
        ! sharded over them inside the library, every other command works as with one device
        block
            character(len=32) :: env_
            integer :: len_, stat_, ndev_
            call get_environment_variable( 'KIWI_HIP_NDEV', env_, len_, stat_ )
            if (stat_ == 0 .and. len_ > 0) then
                read (env_(1:len_), *, iostat=stat_) ndev_
                if (stat_ /= 0) ndev_ = 1
                rc_ = kiwi_hip_init_multi( int(ndev_, c_int), ctx )
            else
                rc_ = kiwi_hip_init( 0_c_int, ctx )
            end if
        end block
        if (rc_ /= 0) then
            errstr = kiwi_hip_error_message( c_null_ptr )
            need_ctx = .false.
            return
        end if
        have_ctx = .true.
    end function

    integer function source_type_id( name )
        character(len=*), intent(in) :: name
        select case (name)                     ! source_all.f90:88-98 / parameterized_source.f90:45-50
        case ('bilateral');     source_type_id = 1
        case ('circular');      source_type_id = 2
        case ('point_lp');      source_type_id = 3
        case ('eikonal');       source_type_id = 4
        case ('mt_eikonal');    source_type_id = 5
        case ('moment_tensor'); source_type_id = 6
        case default;           source_type_id = 0
        end select
    end function

    integer function norm_id( name )
        character(len=*), intent(in) :: name
        select case (name)                     ! comparator.f90:137-146
        case ('l2norm');          norm_id = 1
        case ('l1norm');          norm_id = 2
        case ('ampspec_l2norm');  norm_id = 3
        case ('ampspec_l1norm');  norm_id = 4
        case ('scalar_product');  norm_id = 5
        case ('peak');            norm_id = 6
        case ('floating_l2norm'); norm_id = 7
        case ('floating_l1norm'); norm_id = 8
        case default;             norm_id = 0
        end select
    end function

    character function component_char( s, k )
        character(len=*), intent(in) :: s
        integer, intent(in) :: k
        component_char = s(k:k)
    end function

  ! ------------------------------------------------------------------------------------------
    subroutine do_command( cmd, a, ok_ )
        character(len=*), intent(in) :: cmd, a
        logical, intent(out) :: ok_
        ok_ = .false.
        select case (cmd)
        case ('set_database');              call do_set_database( a, ok_ )
        case ('set_effective_dt');          call do_set_effective_dt( a, ok_ )
        case ('set_local_interpolation');   call do_set_local_interpolation( a, ok_ )
        case ('set_spacial_undersampling'); call do_set_spacial_undersampling( a, ok_ )
        case ('set_receivers');             call do_set_receivers( a, ok_ )
        case ('switch_receiver');           call do_switch_receiver( a, ok_ )
        case ('set_source_location');       call do_set_source_location( a, ok_ )
        case ('set_source_params');         call do_set_source_params( a, ok_ )
        case ('set_source_params_mask');    call do_set_source_params_mask( a, ok_ )
        case ('set_source_subparams');      call do_set_source_subparams( a, ok_ )
        case ('set_source_subparams_limits'); call do_set_source_subparams_limits( a, ok_ )
        case ('get_source_subparams');      call do_get_source_subparams( ok_ )
        case ('minimize_lm');               call do_minimize_lm( ok_ )
        case ('output_cross_correlations'); call do_output_cross_correlations( a, ok_ )
        case ('output_seismogram_spectra'); call do_output_seismogram_spectra( a, ok_ )
        case ('get_principal_axes');        call do_get_principal_axes( ok_ )
        case ('get_peak_amplitudes');       call do_get_shake( a, .true., ok_ )
        case ('get_arias_intensities');     call do_get_shake( a, .false., ok_ )
        case ('set_source_crust');          call do_set_source_crust( a, ok_ )
        case ('set_source_constraints');    call do_set_source_constraints( a, ok_ )
        case ('set_source_crustal_thickness_limit'); call do_set_source_crustal_thickness_limit( a, ok_ )
        case ('get_source_crustal_thickness');       call do_get_source_crustal_thickness( ok_ )
        case ('set_ref_seismograms');       call do_set_ref_seismograms( a, ok_ )
        case ('set_misfit_method');         call do_set_misfit_method( a, ok_ )
        case ('set_misfit_taper');          call do_set_plf( a, .true., .true., ok_ )
        case ('set_misfit_filter');         call do_set_plf( a, .false., .false., ok_ )     ! all receivers, minimizer.f90:875-920
        case ('set_misfit_filter_1');       call do_set_plf( a, .false., .true., ok_ )      ! one receiver, :922-968
        case ('set_synthetics_factor');     call do_set_synthetics_factor( a, ok_ )
        case ('set_floating_shiftrange');   call do_set_floating_shiftrange( a, ok_ )
        case ('shift_ref_seismogram');      call do_shift_ref_seismogram( a, ok_ )
        case ('autoshift_ref_seismogram');  call do_autoshift_ref_seismogram( a, ok_ )
        case ('get_floating_shifts');       call do_get_floating_shifts( ok_ )
        case ('get_misfits');               call do_get_misfits( .false., ok_ )
        case ('get_global_misfit');         call do_get_misfits( .true., ok_ )
        case ('output_seismograms');        call do_output_seismograms( a, ok_ )
        case ('output_distances');          call do_output_distances( a, ok_ )
        case ('output_source_model');       call do_output_source_model( a, ok_ )
        case ('get_cached_traces_memory');  call do_get_cached_traces_memory( ok_ )
        case ('get_database_format')            ! which reader set_database took: "hdf5" (gfdb_io_hdf.f90 layout) or "kiwiflat"
            if (len_trim(db_format) == 0) then
                call fail( 'no database set' )
            else
                answer = trim(db_format); ok_ = .true.
            end if
        case ('set_cached_traces_memory_limit'); ok_ = .true.      ! the database is resident on the device: nothing to limit
        case ('eval_sources');              call do_eval_sources( a, ok_ )
        case ('set_verbose', 'set_ignore_sigint'); ok_ = .true.
        case default
            call fail( 'unknown command: '//cmd )
        end select
    end subroutine

  ! flat dense GFDB: stream file <base>.kiwiflat =
  !   'KIWIFLAT' int32 version(1) nx nz ng L, real32 dt dx dz firstx firstz,
  !   int32 first(ng,nz,nx), int32 nsamp(ng,nz,nx), real32 data(L,ng,nz,nx)
    subroutine do_set_database( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=1024) :: base
        character(len=maxline) :: rest
        character(len=8) :: magic
        integer(c_int) :: version, nx, nz, ng, L
        real(c_float) :: dt, dx, dz, firstx, firstz
        integer(c_int), allocatable :: first(:), nsamp(:)
        real(c_float), allocatable :: G(:)
        integer :: unit, ios
        ok_ = .false.
        call split_first( a, base, rest )
        open( newunit=unit, file=trim(base)//'.kiwiflat', access='stream', form='unformatted', status='old', iostat=ios )
#ifdef HAVE_GFDB_HDF5
        if (ios /= 0) then        ! the reference's own on-disk format: <base>.index + <base>.<i>.chunk (gfdb_io_hdf.f90)
            call set_database_hdf5( trim(base), ok_ )
            return
        end if
#endif
        if (ios /= 0) then
            call fail( "can't open file "//trim(base)//'.kiwiflat' ); return
        end if
        read (unit, iostat=ios) magic, version, nx, nz, ng, L, dt, dx, dz, firstx, firstz
        if (ios /= 0 .or. magic /= 'KIWIFLAT' .or. version /= 1) then
            close( unit ); call fail( 'not a flat kiwi gfdb: '//trim(base)//'.kiwiflat' ); return
        end if
        allocate( first(nx*nz*ng), nsamp(nx*nz*ng), G(int(L,8)*nx*nz*ng) )
        read (unit, iostat=ios) first, nsamp, G
        close( unit )
        if (ios /= 0) then
            call fail( 'truncated gfdb file '//trim(base)//'.kiwiflat' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_gfdb( ctx, nx, nz, ng, L, dt, dx, dz, firstx, firstz, G, first, nsamp ) )
        if (ok_) db_dt = dt
        if (ok_) db_format = 'kiwiflat'
        evaluated = .false.
    end subroutine

#ifdef HAVE_GFDB_HDF5
    subroutine set_database_hdf5( base, ok_ )
        character(len=*), intent(in) :: base
        logical, intent(out) :: ok_
        type, bind(C) :: t_index
            real(c_float) :: dt, dx, dz, firstx, firstz
            integer(c_int) :: nchunks, nx, nxc, nz, ng
        end type
        interface
            integer(c_int) function kiwi_gfdb_read_index( base, ix, err, errlen ) bind(C, name='kiwi_gfdb_read_index')
                import :: c_int, c_char, t_index
                character(kind=c_char), intent(in) :: base(*)
                type(t_index), intent(out) :: ix
                character(kind=c_char), intent(out) :: err(*)
                integer(c_int), value :: errlen
            end function
            integer(c_int) function kiwi_gfdb_read_dense( base, L, G, first, nsamp, lmax, err, errlen ) &
                    bind(C, name='kiwi_gfdb_read_dense')
                import :: c_int, c_char, c_float, c_ptr
                character(kind=c_char), intent(in) :: base(*)
                integer(c_int), value :: L, errlen
                type(c_ptr), value :: G
                integer(c_int), intent(out) :: first(*), nsamp(*), lmax
                character(kind=c_char), intent(out) :: err(*)
            end function
        end interface
        type(t_index) :: ix
        character(kind=c_char) :: err(512)
        character(len=512) :: msg
        integer(c_int), allocatable :: first(:), nsamp(:)
        real(c_float), allocatable, target :: G(:)
        integer(c_int) :: lmax
        integer :: i
        ok_ = .false.
        err = c_null_char
        if (kiwi_gfdb_read_index( base//c_null_char, ix, err, 512_c_int ) /= 0) goto 10
        allocate( first(ix%nx*ix%nz*ix%ng), nsamp(ix%nx*ix%nz*ix%ng) )
        if (kiwi_gfdb_read_dense( base//c_null_char, 0_c_int, c_null_ptr, first, nsamp, lmax, err, 512_c_int ) /= 0) goto 10
        lmax = max(lmax, 1)
        allocate( G(int(lmax,8)*ix%nx*ix%nz*ix%ng) )
        if (kiwi_gfdb_read_dense( base//c_null_char, lmax, c_loc(G), first, nsamp, lmax, err, 512_c_int ) /= 0) goto 10
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_gfdb( ctx, ix%nx, ix%nz, ix%ng, lmax, ix%dt, ix%dx, ix%dz, ix%firstx, ix%firstz, G, first, nsamp ) )
        if (ok_) db_dt = ix%dt
        if (ok_) db_format = 'hdf5'
        evaluated = .false.
        return
10      continue
        msg = ''
        do i=1,512
            if (err(i) == c_null_char) exit
            msg(i:i) = err(i)
        end do
        call fail( trim(msg) )
    end subroutine
#endif

    subroutine do_set_effective_dt( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float) :: v
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) v
        if (ios /= 0) then
            call fail( 'usage: set_effective_dt effective_dt' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_effective_dt( ctx, v ) )
        if (ok_) effective_dt = v
        evaluated = .false.
    end subroutine

    subroutine do_set_local_interpolation( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        ok_ = .false.
        if (a == 'nearest_neighbor') then
            cur_bilinear = 0
        else if (a == 'bilinear') then
            cur_bilinear = 1
        else
            call fail( 'set_local_interpolation: unknown interpolation method: '//a ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_interp( ctx, cur_bilinear, cur_xus, cur_zus ) )
        evaluated = .false.
    end subroutine

    subroutine do_set_spacial_undersampling( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer(c_int) :: xu, zu
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) xu, zu
        if (ios /= 0 .or. xu < 1 .or. zu < 1) then
            call fail( 'set_spacial_undersampling: failed to parse arguments' ); return
        end if
        if (.not. need_ctx()) return
        cur_xus = xu; cur_zus = zu
        ok_ = check( kiwi_hip_set_interp( ctx, cur_bilinear, cur_xus, cur_zus ) )
        evaluated = .false.
    end subroutine

    subroutine do_set_receivers( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=1024) :: fn
        character(len=maxline) :: rest
        character(len=1024) :: str
        logical :: has_depth
        integer :: unit, ios, n, nw, i
        real(c_double), allocatable :: lat(:), lon(:)
        real(c_float), allocatable :: depth(:)
        character(len=8), allocatable :: comps(:)
        character(kind=c_char, len=9), allocatable, target :: cstr(:)
        type(c_ptr), allocatable :: cptr(:)
        ok_ = .false.
        call split_first( a, fn, rest )
        has_depth = (trim(rest) == 'has_depth')
        open( newunit=unit, file=trim(fn), status='old', iostat=ios )
        if (ios /= 0) then
            call fail( "can't open file "//trim(fn) ); return
        end if
        n = 0
        do
            read (unit,'(a)',iostat=ios) str
            if (ios /= 0) exit
            call reduce_whitespace( str )
            if (len_trim(str) > 0) n = n + 1
        end do
        rewind( unit )
        allocate( lat(n), lon(n), depth(n), comps(n), cstr(n), cptr(n) )
        i = 0
        do
            read (unit,'(a)',iostat=ios) str
            if (ios /= 0) exit
            call reduce_whitespace( str )
            if (len_trim(str) == 0) cycle
            i = i + 1
            comps(i) = ''
            depth(i) = 0.
            nw = count_words( str )
            ios = 0
            if (has_depth .and. nw == 4) then
                read (str,*,iostat=ios) lat(i), lon(i), depth(i), comps(i)
            else if (has_depth .and. nw == 3) then
                read (str,*,iostat=ios) lat(i), lon(i), depth(i)
            else if (.not. has_depth .and. nw == 3) then
                read (str,*,iostat=ios) lat(i), lon(i), comps(i)
            else if (.not. has_depth .and. nw == 2) then
                read (str,*,iostat=ios) lat(i), lon(i)
            else
                ios = 1
            end if
            if (ios /= 0) then
                close( unit )
                write (str,'(a,i0,a)') 'expected two or three words at receiver no ', i, ' while reading '//trim(fn)
                call fail( trim(str) ); return
            end if
        end do
        close( unit )
        do i = 1, n
            cstr(i) = trim(comps(i))//c_null_char
            cptr(i) = c_loc( cstr(i) )
        end do
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_receivers( ctx, int(n,c_int), lat, lon, depth, cptr ) )
        if (ok_) then
            nreceivers = n
            if (allocated(components)) deallocate( components, enabled )
            allocate( components(n), enabled(n) )
            components = comps
            do i = 1, n
                enabled(i) = len_trim(comps(i)) > 0
            end do
        end if
        evaluated = .false.
    end subroutine

    subroutine do_switch_receiver( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer(c_int) :: irec
        character(len=16) :: onoff
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) irec, onoff
        if (ios /= 0 .or. (onoff /= 'on' .and. onoff /= 'off')) then
            call fail( 'usage: switch_receiver ireceiver ( on | off )' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_switch_receiver( ctx, irec, merge(1_c_int, 0_c_int, onoff == 'on') ) )
        if (ok_) enabled(irec) = (onoff == 'on')
        evaluated = .false.
    end subroutine

    subroutine do_set_source_location( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float) :: lat, lon
        real(c_double) :: rt
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) lat, lon, rt
        if (ios /= 0) then
            call fail( 'usage: set_source_location latitude longitude reference-time' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_source_location( ctx, lat, lon, rt ) )
        if (ok_) ref_time = rt
        evaluated = .false.
    end subroutine

    ! set_source_crust r1..r31 o1..o31 : the two CRUST2.0 1-D profiles the reference looks up itself in
    ! set_source_location (crust2x2.f90 reads its tables from KIWI_HOME/aux; this host takes the profiles on the wire)
    subroutine do_set_source_crust( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float) :: prof(31,2)
        integer :: ios
        ok_ = .false.
        if (count_words( a ) /= 62) then
            call fail( 'usage: set_source_crust vp(8) vs(8) rho(8) thickness(7) [rupture profile] ... [origin profile]' ); return
        end if
        read (a,*,iostat=ios) prof
        if (ios /= 0) then
            call fail( 'set_source_crust: cannot parse profiles' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_source_crust( ctx, prof(:,1), prof(:,2) ) )
        evaluated = .false.
    end subroutine

    ! set_source_constraints px1 py1 pz1 nx1 ny1 nz1 ...   (minimizer.f90:521-579)
    subroutine do_set_source_constraints( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float), allocatable :: numbers(:,:), points(:,:), normals(:,:)
        integer :: n, ios
        ok_ = .false.
        n = count_words( a )
        if (mod(n,6) /= 0) then
            call fail( 'number of arguments is not divideable by 6' ); return
        end if
        allocate( numbers(6,n/6), points(3,n/6), normals(3,n/6) )
        read (a,*,iostat=ios) numbers
        if (ios /= 0 .and. n > 0) then
            call fail( 'failed to parse constraints' ); return
        end if
        points = numbers(1:3,:)
        normals = numbers(4:6,:)
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_source_constraints( ctx, int(n/6,c_int), points, normals ) )
        evaluated = .false.
    end subroutine

    ! set_source_crustal_thickness_limit thickness-limit   (minimizer.f90:581-611)
    subroutine do_set_source_crustal_thickness_limit( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float) :: limit
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) limit
        if (ios /= 0) then
            call fail( 'usage: set_source_crustal_thickness_limit thickness-limit' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_source_crustal_thickness_limit( ctx, limit ) )
        evaluated = .false.
    end subroutine

    ! get_source_crustal_thickness   (minimizer.f90:613-640)
    subroutine do_get_source_crustal_thickness( ok_ )
        logical, intent(out) :: ok_
        real(c_float) :: thickness
        character(len=64) :: buffer
        ok_ = .false.
        if (.not. need_ctx()) return
        if (.not. check( kiwi_hip_get_source_crustal_thickness( ctx, thickness ) )) return
        write (buffer,*) thickness
        answer = trim(adjustl(buffer))
        ok_ = .true.
    end subroutine

    subroutine do_set_source_params( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=64) :: tname
        character(len=maxline) :: rest
        integer :: st, np, ios
        real(c_float), allocatable :: p(:)
        ok_ = .false.
        call split_first( a, tname, rest )
        st = source_type_id( trim(tname) )
        if (st == 0) then
            call fail( 'unknown source type: '//trim(tname) ); return
        end if
        np = kiwi_hip_source_nparams( int(st,c_int) )
        if (np < 0) then
            call fail( 'source type not available in this host: '//trim(tname) ); return
        end if
        if (count_words( rest ) /= np) then
            write (rest,'(a,i0,a)') 'expected ', np, ' source parameters for source type '//trim(tname)
            call fail( trim(rest) ); return
        end if
        allocate( p(np) )
        read (rest,*,iostat=ios) p
        if (ios /= 0) then
            call fail( 'failed to parse source parameters' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_sources_params( ctx, int(st,c_int), 1_c_int, p ) )
        source_set = ok_
        evaluated = .false.
        if (.not. ok_) return
        if (st /= cur_st .or. .not. allocated(cur_mask)) then        ! psm_set, source_all.f90:249-253: new type, mask all true
            if (allocated(cur_mask)) deallocate( cur_mask )
            allocate( cur_mask(np) )
            cur_mask = 1
        end if
        cur_st = st
        cur_params = p
    end subroutine

  ! set_source_params_mask mask ... (minimizer.f90:694-736, minimizer_engine.f90:525-543): T / F per parameter
    subroutine do_set_source_params_mask( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        logical, allocatable :: m(:)
        integer :: n, ios
        ok_ = .false.
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        n = count_words( a )
        allocate( m(n) )
        read (a,*,iostat=ios) m
        if (ios > 0) then
            call fail( 'failed to parse source parameter mask' ); return
        end if
        if (n /= size(cur_params)) then
            call fail( 'wrong number of elements in mask' ); return
        end if
        cur_mask = merge( 1_c_int, 0_c_int, m )
        if (allocated(sub_mins)) deallocate( sub_mins )            ! reset_subparam_limits
        if (allocated(sub_maxs)) deallocate( sub_maxs )
        ok_ = .true.
    end subroutine

  ! set_source_subparams subparams ... (minimizer.f90:738-770, minimizer_engine.f90:545-566)
    subroutine do_set_source_subparams( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float), allocatable :: sub(:), p(:)
        integer :: n, ios, i, isub
        ok_ = .false.
        n = count_words( a )
        allocate( sub(n) )
        read (a,*,iostat=ios) sub
        if (ios > 0) then
            call fail( 'failed to parse source parameters' ); return
        end if
        if (.not. source_set) then
            call fail( 'source parameters must be set prior to setting parameter subset' ); return
        end if
        if (count(cur_mask /= 0) /= n) then
            call fail( 'wrong number of subparams' ); return
        end if
        p = cur_params
        isub = 1
        do i = 1, size(p)
            if (cur_mask(i) /= 0) then
                p(i) = sub(isub)
                isub = isub + 1
            end if
        end do
        ok_ = check( kiwi_hip_set_sources_params( ctx, cur_st, 1_c_int, p ) )
        evaluated = .false.
        if (ok_) cur_params = p
    end subroutine

  ! set_source_subparams_limits mins ... maxs ... (minimizer.f90:772-810, minimizer_engine.f90:580-611)
    subroutine do_set_source_subparams_limits( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float), allocatable :: temp(:)
        integer :: n, ios
        ok_ = .false.
        n = count_words( a ) / 2
        allocate( temp(2*n) )
        read (a,*,iostat=ios) temp
        if (ios > 0) then
            call fail( 'failed to parse source parameter minimums' ); return
        end if
        if (.not. allocated(cur_mask)) then
            call fail( 'wrong number of subparam_mins' ); return
        end if
        if (count(cur_mask /= 0) /= n) then
            call fail( 'wrong number of subparam_mins' ); return
        end if
        sub_mins = temp(1:n)
        sub_maxs = temp(n+1:2*n)
        ok_ = .true.
    end subroutine

  ! get_source_subparams (minimizer.f90:1199-1224)
    subroutine do_get_source_subparams( ok_ )
        logical, intent(out) :: ok_
        character(len=32) :: buffer
        integer :: i
        ok_ = .false.
        if (.not. source_set) then
            call fail( 'must set source parameters before retrieving them' ); return
        end if
        answer = ''
        do i = 1, size(cur_params)
            if (cur_mask(i) /= 0) then
                write (buffer,*) cur_params(i)
                if (len(answer) > 0) answer = answer//' '
                answer = answer//trim(adjustl(buffer))
            end if
        end do
        ok_ = .true.
    end subroutine

  ! output_seismogram_spectra filenamebase (synthetics|references) (plain|filtered) (minimizer.f90:1102-1149,
  ! receiver.f90:666-708): <base>-<irec>-<comp>.table with lines "frequency [Hz]  amplitude"
    subroutine do_output_seismogram_spectra( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=1024) :: base, fn
        character(len=maxline) :: r1, probe, proc
        integer(c_int) :: which_probe, filtered, n
        real(c_float) :: df
        integer :: irec, k, i, unit, ios
        real(c_float), allocatable :: d(:)
        ok_ = .false.
        call split_first( a, base, r1 )
        call split_first( r1, probe, proc )
        which_probe = 1
        if (trim(probe) == 'references') which_probe = 0
        filtered = 0
        if (trim(proc) == 'filtered') filtered = 1
        if (.not. need_ctx()) return
        if (which_probe == 1 .and. .not. source_set) then
            call fail( 'no source set' ); return
        end if
        allocate( d(1048576) )
        do irec = 1, nreceivers
            if (.not. enabled(irec)) cycle
            do k = 1, len_trim(components(irec))
                if (.not. check( kiwi_hip_get_amp_spectrum( ctx, 0_c_int, int(irec,c_int), int(k,c_int), which_probe, filtered, &
                                                            df, n, d, 1048576_c_int ) )) return
                write (fn,'(a,a,i0,a,a,a)') trim(base), '-', irec, '-', components(irec)(k:k), '.table'
                open( newunit=unit, file=trim(fn), status='unknown', iostat=ios )
                if (ios /= 0) then
                    call fail( 'failed to write output file: '//trim(fn) ); return
                end if
                do i = 1, n
                    write (unit,*) (i - 1) * df, d(i)
                end do
                close( unit )
            end do
        end do
        evaluated = .false.
        ok_ = .true.
    end subroutine

  ! get_principal_axes (minimizer.f90:1374-1402): "p-azimuth p-polar t-azimuth t-polar" in degrees
    subroutine do_get_principal_axes( ok_ )
        logical, intent(out) :: ok_
        real(c_float) :: pax(2), tax(2)
        character(len=128) :: buffer
        ok_ = .false.
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        if (kiwi_hip_principal_axes( cur_st, cur_params, pax, tax ) /= 0) then
            call fail( 'principal axes are defined for bilateral sources only' ); return
        end if
        write (buffer,*) pax(1), pax(2), tax(1), tax(2)
        call reduce_whitespace( buffer )
        answer = trim(adjustl(buffer))
        ok_ = .true.
    end subroutine

  ! output_cross_correlations filenamebase shift-min shift-max (minimizer.f90:1442-1482, minimizer_engine.f90:1283-1306,
  ! receiver.f90:710-744): per enabled receiver and component <base>-<irec>-<comp>.table with lines "shift [s]  value"
    subroutine do_output_cross_correlations( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=1024) :: base, fn
        character(len=maxline) :: rest
        real(c_float) :: smin, smax
        integer(c_int) :: first, ns
        integer :: ios, irec, k, q, unit, nc
        real(c_float), allocatable :: cc(:)
        ok_ = .false.
        call split_first( a, base, rest )
        read (rest,*,iostat=ios) smin, smax
        if (ios /= 0) then
            call fail( 'usage: output_cross_correlations filenamebase shift-min shift-max' ); return
        end if
        if (.not. need_ctx()) return
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        allocate( cc(1048576) )
        do irec = 1, nreceivers
            if (.not. enabled(irec)) cycle
            nc = len_trim(components(irec))
            if (.not. check( kiwi_hip_get_cross_correlations( ctx, 0_c_int, int(irec,c_int), smin, smax, first, ns, cc, &
                                                              1048576_c_int ) )) return
            do k = 1, nc
                write (fn,'(a,a,i0,a,a,a)') trim(base), '-', irec, '-', components(irec)(k:k), '.table'
                open( newunit=unit, file=trim(fn), status='unknown', iostat=ios )
                if (ios /= 0) then
                    call fail( 'failed to write output file: '//trim(fn) ); return
                end if
                do q = 1, ns
                    write (unit,*) real(first,8) * real(db_dt,8) + (q - 1) * db_dt, cc((k-1)*ns + q)
                end do
                close( unit )
            end do
        end do
        evaluated = .false.
        ok_ = .true.
    end subroutine

  ! get_peak_amplitudes differentiate (minimizer.f90:1305-1343) / get_arias_intensities (:1345-1372): one value per
  ! enabled receiver, list-directed
    subroutine do_get_shake( a, peak, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(in) :: peak
        logical, intent(out) :: ok_
        integer :: differentiate, ios, nen, i
        real(c_float), allocatable :: vals(:)
        character(len=:), allocatable :: buffer
        ok_ = .false.
        if (.not. need_ctx()) return
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        nen = count( enabled )
        allocate( vals(max(nen,1)) )
        if (peak) then
            read (a,*,iostat=ios) differentiate
            if (ios /= 0) then
                call fail( 'failed to parse number' ); return
            end if
            if (.not. check( kiwi_hip_get_peak_amplitudes( ctx, 0_c_int, int(differentiate,c_int), vals ) )) return
        else
            if (.not. check( kiwi_hip_get_arias_intensities( ctx, 0_c_int, vals ) )) return
        end if
        allocate( character(len=32*max(nen,1)) :: buffer )
        write (buffer,*) (vals(i), i=1,nen)
        answer = trim(buffer)
        evaluated = .false.
        ok_ = .true.
    end subroutine

  ! minimize_lm (minimizer.f90:1048-1083): answers "info iterations misfit"; the current source afterwards is the one
  ! of the last forward step, as in the reference
    subroutine do_minimize_lm( ok_ )
        logical, intent(out) :: ok_
        integer(c_int) :: info, iterations
        real(c_float) :: misfit_
        real(c_float), allocatable :: best(:)
        character(len=96) :: buffer
        ok_ = .false.
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        allocate( best(size(cur_params)) )
        if (allocated(sub_mins)) then
            rc = kiwi_hip_minimize_lm( ctx, cur_st, cur_params, cur_mask, c_loc(sub_mins), c_loc(sub_maxs), info, iterations, &
                                       misfit_, best )
        else
            rc = kiwi_hip_minimize_lm( ctx, cur_st, cur_params, cur_mask, c_null_ptr, c_null_ptr, info, iterations, misfit_, best )
        end if
        if (.not. check( rc )) return
        evaluated = .false.
        write (buffer,*) info, iterations, misfit_
        call reduce_whitespace( buffer )
        answer = trim(adjustl(buffer))
        ok_ = .true.
    end subroutine

  ! table format: two columns time [s], value (seismogram_io.f90:231-245); first = nint((t0-reftime)/dt)+1
    subroutine do_set_ref_seismograms( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=1024) :: base, fn
        character(len=maxline) :: fmt_
        integer :: irec, k, unit, ios, n, i
        real(c_double) :: t, t0, tlast
        real(c_float) :: v, deltat
        real(c_float), allocatable :: d(:)
        ok_ = .false.
        call split_first( a, base, fmt_ )
        if (trim(fmt_) /= 'table') then
            call fail( 'only the table format is available in this host' ); return
        end if
        if (.not. need_ctx()) return
        do irec = 1, nreceivers
            if (.not. enabled(irec)) cycle
            do k = 1, len_trim(components(irec))
                write (fn,'(a,a,i0,a,a,a)') trim(base), '-', irec, '-', components(irec)(k:k), '.table'
                open( newunit=unit, file=trim(fn), status='old', iostat=ios )
                if (ios /= 0) then
                    call fail( 'failed to read seismogram from file '//trim(fn) ); return
                end if
                n = 0
                do
                    read (unit,*,iostat=ios) t, v
                    if (ios /= 0) exit
                    n = n + 1
                end do
                if (n < 2) then
                    close( unit ); call fail( 'failed to read seismogram from file '//trim(fn) ); return
                end if
                rewind( unit )
                allocate( d(n) )
                do i = 1, n
                    read (unit,*) t, d(i)
                    if (i == 1) t0 = t
                    tlast = t
                end do
                close( unit )
                deltat = real( (tlast - t0) / (n - 1) )
                if (abs(deltat - db_dt) > db_dt / 10000.) then        ! receiver.f90:776-781
                    deallocate( d ); call fail( "sampling rate in file '"//trim(fn)//"' does not match the database" ); return
                end if
                ok_ = check( kiwi_hip_set_reference( ctx, int(irec,c_int), int(k,c_int), &
                                                     int(nint( real(t0 - ref_time) / db_dt ) + 1, c_int), int(n,c_int), d ) )
                deallocate( d )
                if (.not. ok_) return
            end do
        end do
        ok_ = .true.
        evaluated = .false.
    end subroutine

    subroutine do_set_misfit_method( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer :: id
        ok_ = .false.
        id = norm_id( a )
        if (id == 0) then
            call fail( 'unknown norm: '//a ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_misfit_method( ctx, int(id,c_int) ) )
        evaluated = .false.
    end subroutine

    subroutine do_set_plf( a, taper, with_receiver, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(in) :: taper, with_receiver
        logical, intent(out) :: ok_
        integer :: n, i, ios
        integer(c_int) :: irec
        real(c_float), allocatable :: x(:), y(:)
        ok_ = .false.
        irec = 0
        if (with_receiver) then
            n = (count_words( a ) - 1) / 2
            allocate( x(max(n,1)), y(max(n,1)) )
            read (a,*,iostat=ios) irec, (x(i), y(i), i=1,n)
        else
            n = count_words( a ) / 2
            allocate( x(max(n,1)), y(max(n,1)) )
            read (a,*,iostat=ios) (x(i), y(i), i=1,n)
        end if
        if (ios /= 0) then
            call fail( 'failed to parse values' ); return
        end if
        if (.not. need_ctx()) return
        if (taper) then
            ok_ = check( kiwi_hip_set_taper( ctx, irec, int(n,c_int), x, y ) )
        else
            ok_ = check( kiwi_hip_set_filter( ctx, irec, int(n,c_int), x, y ) )
        end if
        evaluated = .false.
    end subroutine

    subroutine do_set_synthetics_factor( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        real(c_float) :: f
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) f
        if (ios /= 0) then
            call fail( 'usage: set_synthetics_factor factor' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_synthetics_factor( ctx, f ) )
        evaluated = .false.
    end subroutine

    ! shift_ref_seismogram ireceiver shift   (minimizer.f90:355-386)
    subroutine do_shift_ref_seismogram( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer(c_int) :: irec
        real(c_float) :: shift
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) irec, shift
        if (ios /= 0) then
            call fail( 'usage: shift_ref_seismogram ireceiver shift' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_shift_ref_seismogram( ctx, irec, shift ) )
        evaluated = .false.
    end subroutine

    ! autoshift_ref_seismogram ireceiver min-shift max-shift   (minimizer.f90:447-486): answers the shifts applied
    subroutine do_autoshift_ref_seismogram( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer(c_int) :: irec
        real(c_float) :: lo, hi
        real(c_float), allocatable :: sh(:)
        character(len=:), allocatable :: buffer
        integer :: ios, n, i
        ok_ = .false.
        read (a,*,iostat=ios) irec, lo, hi
        if (ios /= 0) then
            call fail( 'usage: autoshift_ref_seismogram ireceiver min-shift max-shift' ); return
        end if
        if (.not. need_ctx()) return
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        n = 1
        if (irec == 0) n = size(enabled)
        allocate( sh(max(n,1)) )
        if (.not. check( kiwi_hip_autoshift_ref_seismogram( ctx, irec, lo, hi, 0_c_int, sh ) )) return
        allocate( character(len=32*max(n,1)) :: buffer )
        write (buffer,*) (sh(i), i=1,n)
        answer = trim(buffer)
        evaluated = .false.
        ok_ = .true.
    end subroutine

    ! set_floating_shiftrange ireceiver min-shift max-shift   (minimizer.f90:388-419)
    subroutine do_set_floating_shiftrange( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer(c_int) :: irec
        real(c_float) :: lo, hi
        integer :: ios
        ok_ = .false.
        read (a,*,iostat=ios) irec, lo, hi
        if (ios /= 0) then
            call fail( 'usage: set_floating_shiftrange ireceiver min-shift max-shift' ); return
        end if
        if (.not. need_ctx()) return
        ok_ = check( kiwi_hip_set_floating_shiftrange( ctx, irec, lo, hi ) )
        evaluated = .false.
    end subroutine

    ! get_floating_shifts   (minimizer.f90:421-445): one value per enabled receiver
    subroutine do_get_floating_shifts( ok_ )
        logical, intent(out) :: ok_
        real(c_float), allocatable :: sh(:)
        character(len=:), allocatable :: buffer
        integer :: n, i
        ok_ = .false.
        if (.not. update_misfits()) return
        n = count( enabled )
        allocate( sh(max(n,1)) )
        if (.not. check( kiwi_hip_get_floating_shifts( ctx, 0_c_int, 1_c_int, sh ) )) return
        allocate( character(len=32*max(n,1)) :: buffer )
        write (buffer,*) (sh(i), i=1,n)
        answer = trim(buffer)
        ok_ = .true.
    end subroutine

    logical function update_misfits()
        update_misfits = .false.
        if (.not. need_ctx()) return
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        if (.not. evaluated) then
            if (.not. check( kiwi_hip_eval( ctx, 0_c_int, 1_c_int ) )) return
            evaluated = .true.
        end if
        update_misfits = .true.
    end function

  ! get_misfits: "m11 n11 m12 n12 ..." list-directed (minimizer.f90:1263-1275); get_global_misfit: one value
    subroutine do_get_misfits( global, ok_ )
        logical, intent(in) :: global
        logical, intent(out) :: ok_
        integer(c_int) :: nmis
        real(c_float), allocatable :: m(:), n(:)
        real(c_float) :: g(1)
        character(len=:), allocatable :: buffer
        integer :: i
        ok_ = .false.
        if (.not. update_misfits()) return
        if (.not. check( kiwi_hip_nmisfits( ctx, nmis ) )) return
        allocate( m(nmis), n(nmis) )
        if (.not. check( kiwi_hip_get_misfits( ctx, 0_c_int, 1_c_int, m, n, g ) )) return
        allocate( character(len=64*max(nmis,1)) :: buffer )
        if (global) then
            write (buffer,*) g(1)
        else
            write (buffer,*) (m(i), n(i), i=1,nmis)
        end if
        answer = trim(buffer)
        ok_ = .true.
    end subroutine

    subroutine do_output_seismograms( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=1024) :: base, fn
        character(len=maxline) :: r1, r2, fmt_, probe, proc
        integer(c_int) :: which, first, n
        integer :: irec, k, unit, ios, i
        logical :: refs
        real(c_float), allocatable :: d(:)
        ok_ = .false.
        call split_first( a, base, r1 )
        call split_first( r1, fmt_, r2 )
        call split_first( r2, probe, proc )
        if (trim(fmt_) /= 'table') then
            call fail( 'only the table format is available in this host' ); return
        end if
        refs = (trim(probe) == 'references')
        if (.not. refs .and. trim(probe) /= 'synthetics') then
            call fail( 'unknown probe: '//trim(probe) ); return
        end if
        which = 1
        if (trim(proc) == 'tapered') which = 2
        if (trim(proc) == 'filtered') which = 3
        if (.not. refs .or. which == 3) then
            if (.not. update_misfits()) return
        end if
        allocate( d(1048576) )
        do irec = 1, nreceivers
            if (.not. enabled(irec)) cycle
            do k = 1, len_trim(components(irec))
                if (refs) then
                    if (.not. check( kiwi_hip_get_reference( ctx, int(irec,c_int), int(k,c_int), which, first, n, &
                                                             d, 1048576_c_int ) )) return
                else
                    if (.not. check( kiwi_hip_get_synthetics( ctx, 0_c_int, int(irec,c_int), int(k,c_int), which, first, n, &
                                                              d, 1048576_c_int ) )) return
                end if
                write (fn,'(a,a,i0,a,a,a)') trim(base), '-', irec, '-', components(irec)(k:k), '.table'
                open( newunit=unit, file=trim(fn), status='unknown', iostat=ios )
                if (ios /= 0) then
                    call fail( 'failed to write output file: '//trim(fn) ); return
                end if
                do i = 1, n                  ! receiver.f90:649 + seismogram_io.f90:131-133
                    write (unit,*) ref_time + (first - 1) * real(db_dt,8) + (i - 1) * db_dt, d(i)
                end do
                close( unit )
            end do
        end do
        evaluated = .false.       ! get_synthetics re-evaluates with retained traces
        ok_ = .true.
    end subroutine

  ! dist_deg dist_m azimuth_deg per receiver (minimizer.f90:1430-1432)
    subroutine do_output_distances( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer :: unit, ios, irec
        real(c_double) :: azi, bazi, dist
        real(c_float), parameter :: pi = 3.14159265358979, earthradius = 6371.*1000.
        ok_ = .false.
        if (.not. need_ctx()) return
        open( newunit=unit, file=trim(a), status='unknown', iostat=ios )
        if (ios /= 0) then
            call fail( 'failed to open file for output: '//trim(a) ); return
        end if
        do irec = 1, nreceivers
            if (.not. check( kiwi_hip_get_receiver_geometry( ctx, int(irec,c_int), azi, bazi, dist ) )) then
                close( unit ); return
            end if
            write (unit,*) 360./2./pi*(dist/earthradius), dist, 360./2./pi*azi
        end do
        close( unit )
        ok_ = .true.
    end subroutine

  ! output_source_model filenamebase (minimizer.f90:1226-1262, minimizer_engine.f90:947-977): <base>-tdsm.info and the
  ! centroid table <base>-dsm.table (north east depth time mxx myy mzz mxy mxz myz)
    subroutine do_output_source_model( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        integer(c_int) :: nc
        real(c_float), allocatable :: cent(:,:)
        real(c_float) :: dummy(10,1)
        integer :: unit, ios, i
        ok_ = .false.
        if (.not. need_ctx()) return
        if (.not. source_set) then
            call fail( 'no source set' ); return
        end if
        if (.not. check( kiwi_hip_get_source_centroids( ctx, 0_c_int, 0_c_int, nc, dummy ) )) return
        allocate( cent(10,max(nc,1)) )
        if (.not. check( kiwi_hip_get_source_centroids( ctx, 0_c_int, nc, nc, cent ) )) return
        open( newunit=unit, file=trim(a)//'-tdsm.info', status='unknown', iostat=ios )
        if (ios /= 0) then
            call fail( 'failed to open output file: '//trim(a)//'-tdsm.info' ); return
        end if
        write (unit,'(a)') 'ncentroids'
        write (unit,*) nc
        write (unit,*)
        close( unit )
        open( newunit=unit, file=trim(a)//'-dsm.table', status='unknown', iostat=ios )
        if (ios /= 0) then
            call fail( 'failed to open output file: '//trim(a)//'-dsm.table' ); return
        end if
        do i = 1, nc
            write (unit,'(10(1x,es15.8e2))') cent(:,i)     ! one record per centroid, nine significant digits
        end do
        close( unit )
        ok_ = .true.
    end subroutine

  ! get_cached_traces_memory (minimizer.f90:1141-1163): here the bytes the engine holds on the device
    subroutine do_get_cached_traces_memory( ok_ )
        logical, intent(out) :: ok_
        integer(c_long_long) :: bytes
        character(len=32) :: buffer
        ok_ = .false.
        if (.not. need_ctx()) return
        if (.not. check( kiwi_hip_get_device_bytes( ctx, bytes ) )) return
        write (buffer,'(i0)') bytes
        answer = trim(buffer)
        ok_ = .true.
    end subroutine

  ! batch extension: a whole trial grid in one command
    subroutine do_eval_sources( a, ok_ )
        character(len=*), intent(in) :: a
        logical, intent(out) :: ok_
        character(len=64) :: tname
        character(len=1024) :: pfile, ofile
        character(len=maxline) :: r1, r2, str
        integer :: st, np, unit, ios, nsrc, i, s
        integer(c_int) :: nmis
        integer(c_int), allocatable :: status(:)
        real(c_float), allocatable :: p(:), m(:), n(:), g(:)
        ok_ = .false.
        call split_first( a, tname, r1 )
        call split_first( r1, pfile, r2 )
        ofile = trim(r2)
        st = source_type_id( trim(tname) )
        np = -1
        if (st /= 0) np = kiwi_hip_source_nparams( int(st,c_int) )
        if (np < 0) then
            call fail( 'source type not available in this host: '//trim(tname) ); return
        end if
        open( newunit=unit, file=trim(pfile), status='old', iostat=ios )
        if (ios /= 0) then
            call fail( "can't open file "//trim(pfile) ); return
        end if
        nsrc = 0
        do
            read (unit,'(a)',iostat=ios) str
            if (ios /= 0) exit
            call reduce_whitespace( str )
            if (len_trim(str) > 0) nsrc = nsrc + 1
        end do
        rewind( unit )
        if (nsrc == 0) then
            close( unit ); call fail( 'no sources in '//trim(pfile) ); return
        end if
        allocate( p(np*nsrc) )
        s = 0
        do
            read (unit,'(a)',iostat=ios) str
            if (ios /= 0) exit
            call reduce_whitespace( str )
            if (len_trim(str) == 0) cycle
            read (str,*,iostat=ios) p(s*np+1:s*np+np)
            if (ios /= 0) then
                close( unit ); call fail( 'failed to parse source parameters in '//trim(pfile) ); return
            end if
            s = s + 1
        end do
        close( unit )
        if (.not. need_ctx()) return
      ! trial sources the discretiser rejects do not fail the sweep (seismosizer.py:703-720): they are listed in the
      ! answer and their rows are zeros.  One call for the whole list: the library cuts it into pieces and discretises one
      ! piece on the host while the device evaluates another (kiwi_hip_misfits_for_params); it leaves the engine with the
      ! head of the list, so source 1 of the file is the current source afterwards.  A list of ONE source that cannot be
      ! discretised is the reference's `set_source_params: nok >`.
        allocate( status(nsrc) )
        if (.not. check( kiwi_hip_nmisfits( ctx, nmis ) )) return
        allocate( m(nmis*nsrc), n(nmis*nsrc), g(nsrc) )
        if (.not. check( kiwi_hip_misfits_for_params( ctx, int(st,c_int), int(nsrc,c_int), p, 0_c_int, m, n, g, status ) )) return
        if (nsrc == 1 .and. status(1) /= 0) then
            rc = kiwi_hip_set_sources_params( ctx, int(st,c_int), 1_c_int, p )      ! for its message
            ok_ = check( rc ); return
        end if
        ! (a list none of whose trials could be discretised uploads nothing: the context keeps whatever it held before)
        if (any(status == 0)) source_set = .true.
        open( newunit=unit, file=trim(ofile), status='unknown', iostat=ios )
        if (ios /= 0) then
            call fail( 'failed to open file for output: '//trim(ofile) ); return
        end if
        do s = 0, nsrc - 1
            write (unit,'(*(es16.8e3,1x))') g(s+1), (m(s*nmis+i), n(s*nmis+i), i=1,nmis)
        end do
        close( unit )
        if (any(status == 0)) evaluated = .true.         ! source 0 of the batch is the "current" source
        write (str,'(i0)') nsrc
        answer = trim(str)
        if (any(status /= 0)) then              ! "<nsrc> failed <i1> <i2> ..." (1-based, in trial order)
            answer = answer//' failed'
            do s = 1, nsrc
                if (status(s) /= 0) then
                    write (str,'(i0)') s
                    answer = answer//' '//trim(str)
                end if
            end do
        end if
        ok_ = .true.
    end subroutine

end program
