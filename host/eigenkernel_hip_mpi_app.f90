! eigenkernel_hip_mpi_app.f90 -- the MPI-shaped Fortran host of libek_hip.so: several ranks on a BLACS-style
! process grid, block-cyclic pieces in and out, exactly the data contract solver_main's callers have.
!
!   mpiexec -np P eigenkernel_hip_mpi_app -s <hip|hip_select|general_hip|general_hip_select>
!           --synthetic <N> [--seed-a 1] [--seed-b 2] [-n <n_vec>] [--block-size <NB>]
!           [--comm hook|host|rccl] [--devices <k>] [-c <n_check>] [-o eigenvalues.dat] [-l log.json]
!
! What it mirrors of the reference (read as text, nothing copied):
!   main.f90:29-35, 84-104      mpi_init ... setup_distribution ... eigen_solver ... mpi_finalize, SPMD
!   processes.f90:17-36, 56-65  row-major process grid, layout_procs (largest divisor <= sqrt(P + 1))
!   distribute_matrix.f90:92-148  setup_distributed_matrix incl. the block-size shrink rule and descinit
!   distribute_matrix.f90:401-422 every rank keeps the entries of the global matrix its grid cell owns
!   main.f90:113-118            eigenvalues.dat written by the master
! and of INTEGRATION.md: 1c (--comm hook: the library borrows MPI_Allgatherv, assembles A and B on every
! rank, runs replicated), 1d (--comm rccl: one rank per GPU, id from rank 0 by MPI_Bcast; --comm host: the
! same distributed stages with every exchange through the hook -- the form that runs P ranks on ONE GPU).
! MPI is reached through host/ek_mpi_shim.c (no flang-readable mpi.mod in this image).
module ek_hip_mpi_binding
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
    integer(c_int) function ek_hip_set_allgatherv(fn, user) bind(C, name='ek_hip_set_allgatherv')
      import :: c_int, c_funptr, c_ptr
      type(c_funptr), value :: fn
      type(c_ptr), value :: user
    end function
    integer(c_int) function ek_hip_gather_matrix(m, n, M_loc, desc, nprow, npcol, myrow, mycol, M_full, ldf) &
         bind(C, name='ek_hip_gather_matrix')
      import :: c_int, c_double
      integer(c_int), value :: m, n, nprow, npcol, myrow, mycol, ldf
      real(c_double) :: M_loc(*), M_full(*)
      integer(c_int) :: desc(9)
    end function
    integer(c_int) function ek_hip_comm_attach_host(nranks, rank) bind(C, name='ek_hip_comm_attach_host')
      import :: c_int
      integer(c_int), value :: nranks, rank
    end function
    integer(c_int) function ek_hip_comm_unique_id(id, bytes) bind(C, name='ek_hip_comm_unique_id')
      import :: c_int, c_char
      character(kind=c_char) :: id(*)
      integer(c_int), value :: bytes
    end function
    integer(c_int) function ek_hip_comm_init(id, bytes, nranks, rank) bind(C, name='ek_hip_comm_init')
      import :: c_int, c_char
      character(kind=c_char), intent(in) :: id(*)
      integer(c_int), value :: bytes, nranks, rank
    end function
    integer(c_int) function ek_hip_comm_destroy() bind(C, name='ek_hip_comm_destroy')
      import :: c_int
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
    ! host/ek_mpi_shim.c
    integer(c_int) function ekm_init(rank, nprocs) bind(C, name='ekm_init')
      import :: c_int
      integer(c_int) :: rank, nprocs
    end function
    integer(c_int) function ekm_finalize() bind(C, name='ekm_finalize')
      import :: c_int
    end function
    integer(c_int) function ekm_barrier() bind(C, name='ekm_barrier')
      import :: c_int
    end function
    subroutine ekm_abort(code) bind(C, name='ekm_abort')
      import :: c_int
      integer(c_int), value :: code
    end subroutine
    real(c_double) function ekm_wtime() bind(C, name='ekm_wtime')
      import :: c_double
    end function
    integer(c_int) function ekm_allgatherv(send, count, recv, counts, displs, user) bind(C, name='ekm_allgatherv')
      import :: c_int, c_double, c_long_long, c_ptr
      real(c_double), intent(in) :: send(*)
      integer(c_long_long), value :: count
      real(c_double) :: recv(*)
      integer(c_long_long), intent(in) :: counts(*), displs(*)
      type(c_ptr), value :: user
    end function
    integer(c_int) function ekm_bcast_bytes(buf, nbytes, root) bind(C, name='ekm_bcast_bytes')
      import :: c_int, c_char
      character(kind=c_char) :: buf(*)
      integer(c_int), value :: nbytes, root
    end function
    integer(c_int) function ekm_bcast_doubles(buf, n, root) bind(C, name='ekm_bcast_doubles')
      import :: c_int, c_double
      real(c_double) :: buf(*)
      integer(c_int), value :: n, root
    end function
    integer(c_int) function ekm_max_int(v) bind(C, name='ekm_max_int')
      import :: c_int
      integer(c_int), value :: v
    end function
    integer(c_int) function ekm_min_int(v) bind(C, name='ekm_min_int')
      import :: c_int
      integer(c_int), value :: v
    end function
    real(c_double) function ekm_max_double(v) bind(C, name='ekm_max_double')
      import :: c_double
      real(c_double), value :: v
    end function
  end interface
end module ek_hip_mpi_binding

program eigenkernel_hip_mpi_app
  use, intrinsic :: iso_c_binding
  use ek_hip_mpi_binding
  implicit none

  character(len=1024) :: arg, solver, out_ev, out_log, comm_mode
  integer :: nargs, iarg, ios
  integer :: synth_n, seed_a, seed_b, n_vec, n_check, block_size, n_devices
  logical :: generalized, is_select
  integer(c_int) :: my_rank, n_procs, info, info_all
  integer :: n_procs_row, n_procs_col, my_proc_row, my_proc_col
  integer :: n, nb, max_nb, local_rows, local_cols, problem, i, j, gi, gj
  integer(c_int), target :: desc_a(9), desc_b(9), desc_z(9), desc_full(9)
  real(c_double), allocatable, target :: a_full(:, :), b_full(:, :), a_loc(:, :), b_loc(:, :)
  real(c_double), allocatable :: z_loc(:, :), z_full(:, :), w(:), w0(:)
  real(c_double) :: stage(EK_HIP_N_STAGES), chk(3), spread, t0, t_setup, t_solve
  character(kind=c_char) :: comm_id(128)
  character(len=40), parameter :: stage_names(EK_HIP_N_STAGES) = [character(len=40) :: &
       'reduce_generalized:pdpotrf', 'reduce_generalized:pdsygst', &
       'eigen_solver_scalapack_all:pdsytrd', 'eigen_solver_scalapack_all:gather1', &
       'eigen_solver_scalapack_all:pdstedc', 'eigen_solver_scalapack_all:pdormtr', &
       'recovery_generalized', 'ek_hip:host_device_copies']

  info = ekm_init(my_rank, n_procs)                                   ! main.f90:29-35
  if (info /= 0) stop 'mpi_init failed'

  solver = ''; out_ev = 'eigenvalues.dat'; out_log = 'log.json'; comm_mode = 'hook'
  synth_n = 0; seed_a = 1; seed_b = 2; n_vec = -1; n_check = 0; block_size = 0; n_devices = 1
  nargs = command_argument_count()
  iarg = 1
  do while (iarg <= nargs)
    call get_command_argument(iarg, arg)
    select case (trim(arg))
    case ('-s'); call next_arg(solver)
    case ('-n'); call next_int(n_vec)
    case ('-c'); call next_int(n_check)
    case ('-o'); call next_arg(out_ev)
    case ('-l'); call next_arg(out_log)
    case ('--block-size'); call next_int(block_size)
    case ('--synthetic'); call next_int(synth_n)
    case ('--seed-a'); call next_int(seed_a)
    case ('--seed-b'); call next_int(seed_b)
    case ('--comm'); call next_arg(comm_mode)
    case ('--devices'); call next_int(n_devices)
    case default
      call die('unknown option '//trim(arg), 1)
    end select
    iarg = iarg + 1
  end do
  if (synth_n <= 0) call die('--synthetic <N> is required (this host generates its input)', 1)
  generalized = (index(solver, 'general_') == 1)
  select case (trim(solver))
  case ('hip', 'hip_select', 'general_hip', 'general_hip_select')
  case default
    call die('eigen_solver: Unknown solver '//trim(solver), 1)        ! solver_main.f90:98
  end select
  is_select = (index(solver, '_select') > 0)
  if (n_vec >= 0 .and. .not. is_select) call die('-n is only legal for *_select solvers', 1)
  select case (trim(comm_mode))
  case ('hook', 'host', 'rccl')
  case default
    call die('--comm expects hook, host or rccl', 1)
  end select
  n = synth_n
  if (n_vec < 0) n_vec = n
  if (n_vec > n) call die('-n exceeds the matrix dimension', 1)
  if (n_check < 0 .or. n_check > n_vec) n_check = n_vec
  if (n_devices < 1) n_devices = 1
  problem = 0
  if (generalized) problem = 1

  ! setup_distribution (processes.f90:17-36): row-major grid, rank r at (r / npcol, mod(r, npcol))
  t0 = ekm_wtime()
  call layout_procs(int(n_procs), n_procs_row, n_procs_col)
  my_proc_row = my_rank / n_procs_col
  my_proc_col = mod(int(my_rank), n_procs_col)
  if (my_rank == 0) print '("BLACS process grid: ", I0, " x ", I0, " (", I0, ")")', n_procs_row, n_procs_col, n_procs

  ! setup_distributed_matrix (distribute_matrix.f90:92-148): block size with the shrink rule, descriptor
  nb = 64
  if (block_size > 0) nb = block_size
  max_nb = max(min(n / n_procs_row, n / n_procs_col), 1)
  if (nb > max_nb) then
    if (my_rank == 0) print '("[Warning] setup_distributed_matrix: size of matrix is very small relative to the number of processes")'
    nb = max_nb
  end if
  local_rows = max(1, numroc(n, nb, my_proc_row, n_procs_row))
  local_cols = max(1, numroc(n, nb, my_proc_col, n_procs_col))
  desc_a = [1, 0, n, n, nb, nb, 0, 0, local_rows]
  desc_b = desc_a; desc_z = desc_a
  desc_full = [1, 0, n, n, nb, nb, 0, 0, max(1, n)]
  if (my_rank == 0) print '("Creating distributed matrix A with M, N, MB, NB: ", I0, ", ", I0, ", ", I0, ", ", I0)', &
       n, n, nb, nb

  info = ek_hip_init(int(mod(int(my_rank), n_devices), c_int))
  if (ekm_min_int(info) /= 0) call die('ek_hip_init failed (no GPU? there is no CPU fallback)', int(ekm_min_int(info)))

  ! the input: every rank generates the global matrices (device generator of SURVEY.md 8(d)) and keeps
  ! the entries of its grid cell (distribute_global_sparse_matrix, distribute_matrix.f90:401-422)
  call synthetic_matrix(n, seed_a, a_full)
  allocate (a_loc(local_rows, local_cols), z_loc(local_rows, local_cols), w(n))
  call cut_local(a_full, a_loc)
  if (generalized) then
    call synthetic_matrix(n, seed_b, b_full)
    allocate (b_loc(local_rows, local_cols))
    call cut_local(b_full, b_loc)
  end if
  z_loc = 0.0d0
  w = 0.0d0

  ! the communication layer the library borrows (INTEGRATION.md 1c), and the communicator (1d)
  info = ek_hip_set_allgatherv(c_funloc(ekm_allgatherv), c_null_ptr)
  if (info /= 0) call die_local('ek_hip_set_allgatherv failed', int(info))
  select case (trim(comm_mode))
  case ('host')
    info = ek_hip_comm_attach_host(n_procs, my_rank)
    if (ekm_min_int(info) /= 0) call die('ek_hip_comm_attach_host failed', int(info))
  case ('rccl')
    comm_id = c_null_char
    info = 0
    if (my_rank == 0) info = ek_hip_comm_unique_id(comm_id, 128_c_int)
    if (ekm_min_int(info) /= 0) call die('ek_hip_comm_unique_id failed', int(info))
    info = ekm_bcast_bytes(comm_id, 128_c_int, 0_c_int)
    info = ek_hip_comm_init(comm_id, 128_c_int, n_procs, my_rank)
    if (ekm_min_int(info) /= 0) call die('ek_hip_comm_init failed', int(info))
  end select
  info = ekm_barrier()
  t_setup = ekm_wtime() - t0

  ! eigen_solver (main.f90:100-104): collective, one call per rank
  t0 = ekm_wtime()
  if (generalized) then
    info = ek_hip_solve(problem, n, n_vec, a_loc, desc_a, c_loc(b_loc), c_loc(desc_b), w, z_loc, desc_z, &
         n_procs_row, n_procs_col, my_proc_row, my_proc_col, stage, EK_HIP_N_STAGES)
  else
    info = ek_hip_solve(problem, n, n_vec, a_loc, desc_a, c_null_ptr, c_null_ptr, w, z_loc, desc_z, &
         n_procs_row, n_procs_col, my_proc_row, my_proc_col, stage, EK_HIP_N_STAGES)
  end if
  info_all = ekm_min_int(info)
  if (info_all == 0) info_all = ekm_max_int(info)
  t_solve = ekm_wtime() - t0
  if (info_all /= 0) then
    if (my_rank == 0) print '("info(ek_hip_solve): ", I0)', info_all
    call die('eigen_solver: ek_hip_solve failed', int(info_all))
  end if

  ! every rank holds all eigenvalues: largest difference to rank 0's
  allocate (w0(n))
  w0 = w
  info = ekm_bcast_doubles(w0, int(n_vec, c_int), 0_c_int)
  spread = ekm_max_double(maxval(abs(w(1:n_vec) - w0(1:n_vec))))
  if (my_rank == 0) then
    print '("eigenvalue spread across ranks: ", E12.4)', spread
    open (unit=21, file=trim(out_ev), status='replace', action='write')      ! main.f90:113-118
    do i = 1, n_vec
      write (21, '(I8, " ", E26.16e3)') i, w(i)
    end do
    close (21)
  end if

  ! -c: the pieces of Z back together (the exchange step alone), residual norms by the master against the
  ! ORIGINAL matrices (verifier.f90:75-204 on the device)
  if (n_check > 0) then
    allocate (z_full(n, n))
    z_full = 0.0d0
    info = ek_hip_gather_matrix(int(n, c_int), int(n, c_int), z_loc, desc_z, n_procs_row, n_procs_col, &
         my_proc_row, my_proc_col, z_full, int(n, c_int))
    if (ekm_min_int(info) /= 0) call die('ek_hip_gather_matrix failed', int(info))
    if (my_rank == 0) then
      if (generalized) then
        info = ek_hip_check(0, problem, n, n_check, 1, 1, c_loc(a_full), c_loc(desc_full), c_loc(b_full), &
             c_loc(desc_full), w, z_full, desc_full, chk)
      else
        info = ek_hip_check(0, problem, n, n_check, 1, 1, c_loc(a_full), c_loc(desc_full), c_null_ptr, &
             c_null_ptr, w, z_full, desc_full, chk)
      end if
      if (info /= 0) call die_local('eval_residual_norm failed', int(info))
      print '("A norm: ", E26.16e3)', chk(1)
      print '("residual norm (average): ", E26.16e3)', chk(2)
      print '("residual norm (max): ", E26.16e3)', chk(3)
    end if
  end if

  if (my_rank == 0) call write_log()
  if (trim(comm_mode) /= 'hook') info = ek_hip_comm_destroy()
  info = ekm_barrier()
  info = ekm_finalize()

contains

  subroutine layout_procs(np, np_row, np_col)                        ! processes.f90:56-65
    integer, intent(in) :: np
    integer, intent(out) :: np_row, np_col
    np_row = int(sqrt(dble(np + 1)))
    do while (mod(np, np_row) /= 0)
      np_row = np_row - 1
    end do
    np_col = np / np_row
  end subroutine layout_procs

  integer function numroc(nn, bs, iproc, nprocs)                     ! ScaLAPACK's NUMROC with source process 0
    integer, intent(in) :: nn, bs, iproc, nprocs
    integer :: nblocks, extra
    nblocks = nn / bs
    numroc = (nblocks / nprocs) * bs
    extra = mod(nblocks, nprocs)
    if (iproc < extra) then
      numroc = numroc + bs
    else if (iproc == extra) then
      numroc = numroc + mod(nn, bs)
    end if
  end function numroc

  subroutine cut_local(full, loc)
    real(c_double), intent(in) :: full(:, :)
    real(c_double), intent(out) :: loc(:, :)
    integer :: lr, lc, li, lj
    loc = 0.0d0
    lr = numroc(n, nb, my_proc_row, n_procs_row)
    lc = numroc(n, nb, my_proc_col, n_procs_col)
    do lj = 0, lc - 1
      gj = ((lj / nb) * n_procs_col + my_proc_col) * nb + mod(lj, nb)
      do li = 0, lr - 1
        gi = ((li / nb) * n_procs_row + my_proc_row) * nb + mod(li, nb)
        loc(li + 1, lj + 1) = full(gi + 1, gj + 1)
      end do
    end do
  end subroutine cut_local

  subroutine synthetic_matrix(dim, seed, mat)
    integer, intent(in) :: dim, seed
    real(c_double), allocatable, target, intent(out) :: mat(:, :)
    type(c_ptr) :: dptr
    integer(c_int) :: rc
    allocate (mat(dim, dim))
    rc = ek_hip_malloc(dptr, int(dim, c_long_long) * int(dim, c_long_long) * 8_c_long_long)
    if (rc /= 0) call die_local('ek_hip_malloc failed', int(rc))
    rc = ek_hip_synth_matrix_device(int(dim, c_int), int(seed, c_long_long), dptr, int(dim, c_int))
    if (rc /= 0) call die_local('ek_hip_synth_matrix_device failed', int(rc))
    rc = ek_hip_memcpy_d2h(mat, dptr, int(dim, c_long_long) * int(dim, c_long_long) * 8_c_long_long)
    if (rc /= 0) call die_local('ek_hip_memcpy_d2h failed', int(rc))
    rc = ek_hip_free(dptr)
  end subroutine synthetic_matrix

  subroutine next_arg(val)
    character(len=*), intent(out) :: val
    iarg = iarg + 1
    if (iarg > nargs) call die('missing value for '//trim(arg), 1)
    call get_command_argument(iarg, val)
  end subroutine next_arg

  subroutine next_int(val)
    integer, intent(out) :: val
    character(len=64) :: sval
    call next_arg(sval)
    read (sval, *, iostat=ios) val
    if (ios /= 0) call die('integer expected after '//trim(arg), 1)
  end subroutine next_int

  ! Collective: every rank calls it with the same error (argument errors, agreed return codes).  The master
  ! says why (processes.f90:133-138), then all ranks end.
  subroutine die(msg, code)
    character(len=*), intent(in) :: msg
    integer, intent(in) :: code
    integer(c_int) :: rc
    if (my_rank == 0) then
      write (0, '("[Error] ", A)') msg
      flush (0)
    end if
    flush (6)
    rc = ekm_barrier()
    call ekm_abort(int(max(1, abs(code)), c_int))
    stop 1
  end subroutine die

  ! One rank alone has failed: it says so itself and takes the job down.
  subroutine die_local(msg, code)
    character(len=*), intent(in) :: msg
    integer, intent(in) :: code
    write (0, '("[Error] rank ", I0, ": ", A)') my_rank, msg
    flush (0)
    call ekm_abort(int(max(1, abs(code)), c_int))
    stop 1
  end subroutine die_local

  subroutine write_log()
    integer :: k
    open (unit=23, file=trim(out_log), status='replace', action='write')
    write (23, '(A)') '{'
    write (23, '(A, I0, A)') '  "n_procs": ', n_procs, ','
    write (23, '(A, I0, A, I0, A)') '  "grid": [', n_procs_row, ', ', n_procs_col, '],'
    write (23, '(A, I0, A)') '  "block_size": ', nb, ','
    write (23, '(A, A, A)') '  "solver": "', trim(solver), '",'
    write (23, '(A, A, A)') '  "comm": "', trim(comm_mode), '",'
    write (23, '(A, I0, A)') '  "n": ', n, ','
    write (23, '(A, I0, A)') '  "n_vec": ', n_vec, ','
    write (23, '(A)') '  "events": ['
    write (23, '(A, E16.8, A)') '    {"name": "main:setup", "val": ', t_setup, '},'
    do k = 1, EK_HIP_N_STAGES
      write (23, '(A, A, A, E16.8, A)') '    {"name": "', trim(stage_names(k)), '", "val": ', stage(k), '},'
    end do
    write (23, '(A, E16.8, A)') '    {"name": "eigen_solver", "val": ', t_solve, '}'
    write (23, '(A)') '  ]'
    write (23, '(A)') '}'
    close (23)
  end subroutine write_log

end program eigenkernel_hip_mpi_app
