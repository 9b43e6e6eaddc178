! eigenkernel_hip_app -- a minimal Fortran host with EigenKernel's command line and file
! contract (README.md:61-98 of the reference; SURVEY.md App. C), driving libek_hip.so through
! ISO_C_BINDING exactly as INTEGRATION.md describes.  Single process = the 1x1 grid.
!
!   eigenkernel_hip_app -s <hip|hip_select|general_hip|general_hip_select> [options] A.mtx [B.mtx]
!     -n <num>    eigenpairs to compute (*_select only; default all)      command_argument.f90:186-200
!     -c <num>    residual check on the first <num> vectors (-1 = all)   main.f90:149-160
!     -t <a>,<b>  orthogonality check on vectors a..b                    main.f90:165-179
!     -o <file>   eigenvalue file (default eigenvalues.dat)              main.f90:111-121
!     -i <file>   IPR file (default ipratios.dat)                        main.f90:131-143
!     -l <file>   log file (default log.json)                            main.f90:185-190
!     -p <ranges> eigenvectors to print, e.g. 1-3,7                      command_argument.f90:271-316
!     -d <dir>    directory of the eigenvector files <dir>/%08d.dat      matrix_io.f90:173-231
!     --binary    one unformatted sequential record of N doubles per file (else `i j value` lines)
!     --block-size <nb>   block size recorded in the descriptors
!     --dry-run   read the matrices and stop                             main.f90:89-93
!     --synthetic <N> [--seed-a 1] [--seed-b 2]   no matrix files: the dense SPD test matrices of
!                 SURVEY.md 8(d) (the generator of bench.py and of the oracle), made on the GPU by the
!                 library and fetched into the host arrays; general_* solvers get the pair (A, B).
!                 Stands where the reference reads + scatters a MatrixMarket file
!                 (matrix_io.f90:91-144, distribute_matrix.f90:401-422): 4.5 GB of text at N = 16384.
!
! Written from scratch for this repository; it mirrors behaviour, not code.
module ek_hip_binding
  use, intrinsic :: iso_c_binding
  implicit none
  integer, parameter :: EK_HIP_N_STAGES = 8
  interface
    integer(c_int) function ek_hip_init(device) bind(C, name='ek_hip_init')
      import :: c_int
      integer(c_int), value :: device
    end function
    integer(c_int) function ek_hip_solve(problem, n, n_vec, A_loc, desc_A, B_loc, desc_B, w, Z_loc, &
         desc_Z, nprow, npcol, myrow, mycol, stage_seconds, n_stages) bind(C, name='ek_hip_solve')
      import :: c_int, c_double, c_ptr
      integer(c_int), value :: problem, n, n_vec, nprow, npcol, myrow, mycol, n_stages
      real(c_double) :: A_loc(*), w(*), Z_loc(*), stage_seconds(*)
      type(c_ptr), value :: B_loc, desc_B
      integer(c_int) :: desc_A(9), desc_Z(9)
    end function
    integer(c_int) function ek_hip_malloc(dptr, bytes) bind(C, name='ek_hip_malloc')
      import :: c_int, c_ptr, c_long_long
      type(c_ptr) :: dptr
      integer(c_long_long), value :: bytes
    end function
    integer(c_int) function ek_hip_free(dptr) bind(C, name='ek_hip_free')
      import :: c_int, c_ptr
      type(c_ptr), value :: dptr
    end function
    integer(c_int) function ek_hip_memcpy_d2h(dst, src, bytes) bind(C, name='ek_hip_memcpy_d2h')
      import :: c_int, c_ptr, c_long_long, c_double
      real(c_double) :: dst(*)
      type(c_ptr), value :: src
      integer(c_long_long), value :: bytes
    end function
    integer(c_int) function ek_hip_synth_matrix_device(n, seed, dM, ldm) bind(C, name='ek_hip_synth_matrix_device')
      import :: c_int, c_ptr, c_long_long
      integer(c_int), value :: n, ldm
      integer(c_long_long), value :: seed
      type(c_ptr), value :: dM
    end function
    integer(c_int) function ek_hip_check(what, problem, n, n_cols, index1, index2, A_loc, desc_A, B_loc, &
         desc_B, w, Z_loc, desc_Z, res) bind(C, name='ek_hip_check')
      import :: c_int, c_double, c_ptr
      integer(c_int), value :: what, problem, n, n_cols, index1, index2
      type(c_ptr), value :: A_loc, desc_A, B_loc, desc_B
      real(c_double) :: w(*), Z_loc(*), res(*)
      integer(c_int) :: desc_Z(9)
    end function
  end interface
end module ek_hip_binding

program eigenkernel_hip_app
  use, intrinsic :: iso_c_binding
  use ek_hip_binding
  implicit none

  character(len=1024) :: arg, solver, file_a, file_b, out_ev, out_ipr, out_log, cmdline, vec_dir, vec_ranges
  logical :: binary_out
  integer, parameter :: max_ranges = 100
  integer :: n_ranges, ranges(2, max_ranges), ir
  integer :: nargs, iarg, n_files, n_vec, n_check, ortho_a, ortho_b, block_size, ios, comma
  logical :: dry_run, generalized, is_select
  integer :: synth_n, seed_a, seed_b
  integer :: n, nb, info, problem, i, j
  integer(c_int), target :: desc_a(9), desc_b(9), desc_z(9)
  real(c_double), allocatable, target :: a_work(:, :), b_work(:, :), a_orig(:, :), b_orig(:, :)
  real(c_double), allocatable :: z(:, :), w(:), ipr(:)
  real(c_double) :: stage(EK_HIP_N_STAGES), chk(3)
  real(c_double), allocatable :: chk_ipr(:)
  integer(8) :: clock0, clock1, clock_rate
  double precision :: t_read, t_solve, t_total
  character(len=40), parameter :: stage_names(EK_HIP_N_STAGES) = [character(len=40) :: &
       'reduce_generalized:pdpotrf', 'reduce_generalized:pdsygst', &
       'eigen_solver_scalapack_all:pdsytrd', 'eigen_solver_scalapack_all:gather1', &
       'eigen_solver_scalapack_all:pdstedc', 'eigen_solver_scalapack_all:pdormtr', &
       'recovery_generalized', 'ek_hip:host_device_copies']
  integer, parameter :: max_events = 32
  character(len=64) :: ev_name(max_events)
  double precision :: ev_val(max_events)
  integer :: n_events

  solver = ''; file_a = ''; file_b = ''
  out_ev = 'eigenvalues.dat'; out_ipr = 'ipratios.dat'; out_log = 'log.json'
  n_vec = -1; n_check = 0; ortho_a = 0; ortho_b = 0; block_size = 0
  dry_run = .false.; n_files = 0; n_events = 0
  synth_n = 0; seed_a = 1; seed_b = 2
  vec_dir = '.'; vec_ranges = ''; binary_out = .false.; n_ranges = 0
  call get_command(cmdline)

  nargs = command_argument_count()
  iarg = 1
  do while (iarg <= nargs)
    call get_command_argument(iarg, arg)
    select case (trim(arg))
    case ('-s'); call next_arg(solver)
    case ('-n'); call next_int(n_vec)
    case ('-c'); call next_int(n_check)
    case ('-o'); call next_arg(out_ev)
    case ('-i'); call next_arg(out_ipr)
    case ('-l'); call next_arg(out_log)
    case ('--block-size'); call next_int(block_size)
    case ('--dry-run'); dry_run = .true.
    case ('-p'); call next_arg(vec_ranges)
    case ('-d'); call next_arg(vec_dir)
    case ('--binary'); binary_out = .true.
    case ('--synthetic'); call next_int(synth_n)
    case ('--seed-a'); call next_int(seed_a)
    case ('--seed-b'); call next_int(seed_b)
    case ('-t')
      call next_arg(arg)
      comma = index(arg, ',')
      if (comma == 0) call die('-t expects <a>,<b>', 1)
      read (arg(1:comma - 1), *, iostat=ios) ortho_a
      if (ios /= 0) call die('-t expects <a>,<b>', 1)
      read (arg(comma + 1:), *, iostat=ios) ortho_b
      if (ios /= 0) call die('-t expects <a>,<b>', 1)
    case ('-h')
      call usage()
      stop
    case default
      if (arg(1:1) == '-') call die('unknown option '//trim(arg), 1)
      n_files = n_files + 1
      if (n_files == 1) file_a = arg
      if (n_files == 2) file_b = arg
      if (n_files > 2) call die('at most two matrix files', 1)
    end select
    iarg = iarg + 1
  end do
  if (n_files == 0 .and. synth_n <= 0) call die('no matrix file given', 1)
  if (n_files > 0 .and. synth_n > 0) call die('--synthetic takes no matrix files', 1)
  generalized = (n_files == 2)
  if (synth_n > 0) generalized = (index(solver, 'general_') == 1)
  select case (trim(solver))
  case ('hip', 'hip_select')
    if (generalized) call die('solver '//trim(solver)//' is for the standard problem (one matrix file)', 1)
  case ('general_hip', 'general_hip_select')
    if (.not. generalized) call die('solver '//trim(solver)//' needs two matrix files', 1)
  case default
    call die('eigen_solver: Unknown solver '//trim(solver), 1)       ! solver_main.f90:98
  end select
  is_select = (index(solver, '_select') > 0)
  if (n_vec >= 0 .and. .not. is_select) call die('-n is only legal for *_select solvers', 1)

  call system_clock(clock0, clock_rate)
  if (synth_n > 0) then
    n = synth_n
    call synthetic_matrix(n, seed_a, a_orig)
    if (generalized) call synthetic_matrix(n, seed_b, b_orig)
  else
    call read_matrix_market(trim(file_a), n, a_orig)
    if (generalized) then
      call read_matrix_market(trim(file_b), i, b_orig)
      if (i /= n) call die('matrix dimensions differ', 1)
    end if
  end if
  call system_clock(clock1)
  t_read = dble(clock1 - clock0) / dble(clock_rate)
  call log_event('main:read_matrix_files', t_read)
  print '("matrix dimension: ", i0)', n
  if (dry_run) then
    print '("dry run: stopping after reading the input")'
    stop
  end if
  if (n_vec < 0) n_vec = n
  if (n_vec > n) call die('-n exceeds the matrix dimension', 1)
  if (n_check < 0 .or. n_check > n_vec) n_check = n_vec
  if (ortho_b > n_vec) call die('-t range exceeds the number of computed vectors', 1)
  if (len_trim(vec_ranges) > 0) call parse_ranges(trim(vec_ranges))
  do ir = 1, n_ranges
    if (ranges(1, ir) < 1 .or. ranges(2, ir) > n_vec .or. ranges(1, ir) > ranges(2, ir)) &
         call die('-p range outside 1..n_vec', 1)      ! command_argument.f90:202-208
  end do

  ! setup_distributed_matrix on the 1x1 grid (distribute_matrix.f90:92-148 incl. the shrink rule)
  nb = 64
  if (block_size > 0) nb = block_size
  if (nb > max(n, 1)) nb = max(n, 1)
  print '("BLACS process grid: 1 x 1 (1)")'
  desc_a = [1, 0, n, n, nb, nb, 0, 0, max(1, n)]
  desc_b = desc_a; desc_z = desc_a
  print '("Creating distributed matrix A with M, N, MB, NB: ", i0, ", ", i0, ", ", i0, ", ", i0)', n, n, nb, nb
  allocate (a_work(n, n), z(n, n), w(n), ipr(n))
  a_work = a_orig
  z = 0.0d0
  problem = 0
  if (generalized) then
    allocate (b_work(n, n))
    b_work = b_orig
    problem = 1
  end if

  info = ek_hip_init(0_c_int)
  if (info /= 0) call die('ek_hip_init failed (no GPU? there is no CPU fallback)', info)
  call system_clock(clock0)
  if (generalized) then
    info = ek_hip_solve(problem, n, n_vec, a_work, desc_a, c_loc(b_work), c_loc(desc_b), w, z, desc_z, &
         1, 1, 0, 0, stage, EK_HIP_N_STAGES)
  else
    info = ek_hip_solve(problem, n, n_vec, a_work, desc_a, c_null_ptr, c_null_ptr, w, z, desc_z, &
         1, 1, 0, 0, stage, EK_HIP_N_STAGES)
  end if
  call system_clock(clock1)
  t_solve = dble(clock1 - clock0) / dble(clock_rate)
  do i = 1, EK_HIP_N_STAGES
    call log_event(trim(stage_names(i)), stage(i))
  end do
  call log_event('eigen_solver', t_solve)
  if (info /= 0) then
    print '("info(ek_hip_solve): ", i0)', info
    call die('eigen_solver: ek_hip_solve failed', info)
  end if

  ! eigenvalues.dat: (I8, " ", E26.16e3) per line (main.f90:113-118)
  open (unit=21, file=trim(out_ev), status='replace', action='write')
  do i = 1, n_vec
    write (21, '(I8, " ", E26.16e3)') i, w(i)
  end do
  close (21)

  ! ipratios.dat: always n lines (main.f90:138-142); columns beyond n_vec were not computed
  allocate (chk_ipr(max(n, 3)))
  chk_ipr = 0.0d0
  if (generalized) then
    info = ek_hip_check(2, problem, n, n_vec, 1, 1, c_null_ptr, c_null_ptr, c_loc(b_orig), c_loc(desc_b), &
         w, z, desc_z, chk_ipr)
  else
    info = ek_hip_check(2, problem, n, n_vec, 1, 1, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, &
         w, z, desc_z, chk_ipr)
  end if
  if (info /= 0) call die('get_ipratios failed', info)
  open (unit=22, file=trim(out_ipr), status='replace', action='write')
  do i = 1, n
    if (i <= n_vec) then
      write (22, '(I8, " ", E26.16e3)') i, chk_ipr(i)
    else
      write (22, '(I8, " ", A26)') i, 'NaN'
    end if
  end do
  close (22)

  ! eigenvector files <dir>/%08d.dat (matrix_io.f90:173-285)
  do ir = 1, n_ranges
    do j = ranges(1, ir), ranges(2, ir)
      call write_eigenvector(j)
    end do
  end do

  if (n_check > 0) then
    if (generalized) then
      info = ek_hip_check(0, problem, n, n_check, 1, 1, c_loc(a_orig), c_loc(desc_a), c_loc(b_orig), &
           c_loc(desc_b), w, z, desc_z, chk)
    else
      info = ek_hip_check(0, problem, n, n_check, 1, 1, c_loc(a_orig), c_loc(desc_a), c_null_ptr, &
           c_null_ptr, w, z, desc_z, chk)
    end if
    if (info /= 0) call die('eval_residual_norm failed', info)
    print '("A norm: ", E26.16e3)', chk(1)
    print '("residual norm (average): ", E26.16e3)', chk(2)
    print '("residual norm (max): ", E26.16e3)', chk(3)
  end if
  if (ortho_a >= 1 .and. ortho_b >= ortho_a) then
    if (generalized) then
      info = ek_hip_check(1, problem, n, 0, ortho_a, ortho_b, c_null_ptr, c_null_ptr, c_loc(b_orig), &
           c_loc(desc_b), w, z, desc_z, chk)
    else
      info = ek_hip_check(1, problem, n, 0, ortho_a, ortho_b, c_null_ptr, c_null_ptr, c_null_ptr, &
           c_null_ptr, w, z, desc_z, chk)
    end if
    if (info /= 0) call die('eval_orthogonality failed', info)
    print '("orthogonality criterion: ", E26.16e3)', chk(1)
  end if

  call system_clock(clock1)
  t_total = t_read + dble(clock1 - clock0) / dble(clock_rate)
  call log_event('main', t_total)
  call write_log()

contains

  ! SURVEY.md 8(d) generator, evaluated by the library on the device (bit-identical to the oracle's and to
  ! bench.py's inputs by construction), fetched into a host array like a matrix read from a file
  subroutine synthetic_matrix(dim, seed, mat)
    integer, intent(in) :: dim, seed
    real(c_double), allocatable, target, intent(out) :: mat(:, :)
    type(c_ptr) :: dptr
    integer(c_long_long) :: bytes
    integer(c_int) :: rc
    if (dim < 1) call die('--synthetic needs a positive order', 1)
    allocate (mat(dim, dim))
    bytes = int(dim, c_long_long) * int(dim, c_long_long) * 8_c_long_long
    rc = ek_hip_init(0_c_int)
    if (rc /= 0) call die('ek_hip_init failed', int(rc))
    rc = ek_hip_malloc(dptr, bytes)
    if (rc /= 0) call die('ek_hip_malloc failed', int(rc))
    rc = ek_hip_synth_matrix_device(int(dim, c_int), int(seed, c_long_long), dptr, int(dim, c_int))
    if (rc /= 0) call die('ek_hip_synth_matrix_device failed', int(rc))
    rc = ek_hip_memcpy_d2h(mat, dptr, bytes)
    if (rc /= 0) call die('ek_hip_memcpy_d2h failed', int(rc))
    rc = ek_hip_free(dptr)
  end subroutine synthetic_matrix

  subroutine next_arg(val)
    character(len=*), intent(out) :: val
    iarg = iarg + 1
    if (iarg > nargs) call die('missing value after option', 1)
    call get_command_argument(iarg, val)
  end subroutine next_arg

  subroutine next_int(val)
    integer, intent(out) :: val
    character(len=64) :: buf
    call next_arg(buf)
    read (buf, *, iostat=ios) val
    if (ios /= 0) call die('integer expected after option', 1)
  end subroutine next_int

  ! "a-b,c,d-e" -> ranges(2, n_ranges)
  subroutine parse_ranges(spec)
    character(len=*), intent(in) :: spec
    integer :: p0, p1, dash, st
    p0 = 1
    do while (p0 <= len(spec))
      p1 = index(spec(p0:), ',')
      if (p1 == 0) then
        p1 = len(spec)
      else
        p1 = p0 + p1 - 2
      end if
      if (p1 < p0) call die('-p: empty range', 1)
      if (n_ranges >= max_ranges) call die('-p: too many ranges', 1)
      n_ranges = n_ranges + 1
      dash = index(spec(p0:p1), '-')
      if (dash == 0) then
        read (spec(p0:p1), *, iostat=st) ranges(1, n_ranges)
        ranges(2, n_ranges) = ranges(1, n_ranges)
      else
        if (dash == 1 .or. p0 + dash - 1 == p1) call die('-p: invalid hyphen placement', 1)
        read (spec(p0:p0 + dash - 2), *, iostat=st) ranges(1, n_ranges)
        if (st == 0) read (spec(p0 + dash:p1), *, iostat=st) ranges(2, n_ranges)
      end if
      if (st /= 0) call die('-p: integer expected', 1)
      p0 = p1 + 2
    end do
  end subroutine parse_ranges

  subroutine write_eigenvector(jvec)
    integer, intent(in) :: jvec
    character(len=1200) :: fname
    integer :: st, irow
    write (fname, '(a, "/", i8.8, ".dat")') trim(vec_dir), jvec
    if (binary_out) then
      open (unit=24, file=trim(fname), form='unformatted', access='sequential', status='replace', iostat=st)
      if (st /= 0) call die('print_eigenvectors: cannot open '//trim(fname), st)
      write (24) z(1:n, jvec)
    else
      open (unit=24, file=trim(fname), status='replace', iostat=st)
      if (st /= 0) call die('print_eigenvectors: cannot open '//trim(fname), st)
      do irow = 1, n
        write (24, "(I8, ' ', I8, ' ', E26.16e3)") irow, jvec, z(irow, jvec)
      end do
    end if
    close (24)
  end subroutine write_eigenvector

  subroutine usage()
    print '(a)', 'usage: eigenkernel_hip_app -s <hip|hip_select|general_hip|general_hip_select> [-n num]'
    print '(a)', '       [-c num] [-t a,b] [-o file] [-i file] [-l file] [-p ranges] [-d dir] [--binary]'
    print '(a)', '       [--block-size nb] [--dry-run] A.mtx [B.mtx]'
  end subroutine usage

  subroutine die(msg, code)
    character(len=*), intent(in) :: msg
    integer, intent(in) :: code
    write (0, '("[Error] ", a)') trim(msg)          ! processes.f90:133-138
    if (code == 0) stop
    error stop 1
  end subroutine die

  subroutine log_event(name, val)
    character(len=*), intent(in) :: name
    double precision, intent(in) :: val
    write (0, '("[Event] ", a, ", ", ES14.6)') trim(name), val
    if (n_events < max_events) then
      n_events = n_events + 1
      ev_name(n_events) = name
      ev_val(n_events) = val
    end if
  end subroutine log_event

  ! MatrixMarket `coordinate real symmetric`, one triangle, mirrored on load
  ! (matrix_io.f90:72-144, distribute_matrix.f90:411-418)
  subroutine read_matrix_market(path, dim, mat)
    character(len=*), intent(in) :: path
    integer, intent(out) :: dim
    real(c_double), allocatable, target, intent(out) :: mat(:, :)
    character(len=1024) :: line
    integer :: rows, cols, nnz, k, ii, jj, st
    double precision :: v
    open (unit=11, file=path, status='old', action='read', iostat=st)
    if (st /= 0) call die('cannot open '//trim(path), 1)
    read (11, '(a)', iostat=st) line
    if (st /= 0) call die('empty matrix file '//trim(path), 1)
    call lower_case(line)
    if (index(line, '%%matrixmarket matrix coordinate real symmetric') /= 1) &
         call die('unsupported MatrixMarket banner in '//trim(path), 1)
    do
      read (11, '(a)', iostat=st) line
      if (st /= 0) call die('missing size line in '//trim(path), 1)
      if (line(1:1) /= '%') exit
    end do
    read (line, *, iostat=st) rows, cols, nnz
    if (st /= 0 .or. rows /= cols) call die('bad size line in '//trim(path), 1)
    dim = rows
    allocate (mat(dim, dim))
    mat = 0.0d0
    do k = 1, nnz
      read (11, *, iostat=st) ii, jj, v
      if (st /= 0) call die('truncated matrix file '//trim(path), 1)
      mat(ii, jj) = v
      mat(jj, ii) = v
    end do
    close (11)
  end subroutine read_matrix_market

  subroutine lower_case(sline)
    character(len=*), intent(inout) :: sline
    integer :: k, c
    do k = 1, len_trim(sline)
      c = iachar(sline(k:k))
      if (c >= iachar('A') .and. c <= iachar('Z')) sline(k:k) = achar(c + 32)
    end do
  end subroutine lower_case

  ! log.json: {"setting": {...9 keys...}, "events": [{"name","num_repeated","val"}, ...]}
  ! (command_argument.f90:494-576, event_logger.f90:104-141; events newest first)
  subroutine write_log()
    integer :: k
    open (unit=23, file=trim(out_log), status='replace', action='write')
    write (23, '(a)') '{"setting": {"version": "ek_hip-1", "command": "'//trim(json_escape(cmdline))//'",'
    write (23, '(a)') ' "matrix_A_filename": "'//trim(json_escape(file_a))//'", "matrix_B_filename": "'// &
         trim(json_escape(file_b))//'",'
    write (23, '(a, i0, a)') ' "log_filename": "'//trim(json_escape(out_log))//'", "dimension": ', n, &
         ', "solver": "'//trim(solver)//'",'
    write (23, '(a, i0, a)') ' "g_block_size": 64, "block_size": ', block_size, '},'
    write (23, '(a)') ' "events": ['
    do k = n_events, 1, -1
      write (23, '(a, ES23.15E3, a)', advance='no') '  {"name": "'//trim(ev_name(k))//'", "num_repeated": 1, "val": ', &
           ev_val(k), '}'
      if (k > 1) then
        write (23, '(a)') ','
      else
        write (23, '(a)') ''
      end if
    end do
    write (23, '(a)') ' ]}'
    close (23)
  end subroutine write_log

  function json_escape(sin) result(sout)
    character(len=*), intent(in) :: sin
    character(len=2 * len(sin)) :: sout
    integer :: k, m
    sout = ''
    m = 0
    do k = 1, len_trim(sin)
      if (sin(k:k) == '"' .or. sin(k:k) == '\') then
        m = m + 1; sout(m:m) = '\'
      end if
      m = m + 1; sout(m:m) = sin(k:k)
    end do
  end function json_escape

end program eigenkernel_hip_app
