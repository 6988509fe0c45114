! forgex -- drop-in Fortran module for the batch match path, MI355X build.
!
! Same public names as the reference module (reference src/forgex.F90:24-54):
!     is_valid_regex, operator(.in.), operator(.match.), regex, regex_f
! with the same argument lists, the same PURITY (`pure elemental` operators and is_valid_regex, `pure` regex / regex_f:
! reference src/forgex.F90:58,74,163,235,351) and the same error behaviour, bound through iso_c_binding to the C ABI of
! include/forgex_amd.h (libforgex_amd.so: host table compiler + HIP kernels for gfx950).
!
! What a maintainer should know (INTEGRATION.md):
!   * The C entry points are bound as PURE SUBROUTINES (the library's fxamd_f_* forms: every output is an INTENT(OUT) argument, which
!     is what F2018 allows a pure procedure -- gfortran and flang both accept it).  That is the contract the library keeps: a call's
!     results depend on its arguments only, nothing the caller can observe is modified, handles are thread-safe -- so the
!     public procedures keep the reference's `pure` / `elemental` attributes and stay callable from pure procedures and
!     `do concurrent`.  (The reference's own IMPURE switch, src/forgex.F90:10-13, is available here too: -DIMPURE.)
!   * Rank-1 character arrays resolve to NON-elemental batch specifics (a non-elemental specific is preferred over an
!     elemental one), so `pattern .in. strs(:)` compiles the pattern ONCE and matches all rows in one kernel launch
!     instead of re-parsing the pattern per element (reference src/forgex.F90:98,139-140); `patterns(:) .in. strs(:)`
!     compiles every DISTINCT pattern once and matches the rows of each in one batch.
!   * `regex` gains a batch form returning from(:)/to(:) for a rank-1 text array.
!   * type(fx_batch) (round 4) keeps a batch RESIDENT in HBM across calls and patterns: `batch = fx_batch_upload(strs)` (or
!     fx_batch_wrap of a device pointer the caller owns), then the SAME operators and `regex` with the batch in place of the character
!     array -- `pattern .in. batch`, `pattern .match. batch`, `patterns(:) .in. batch`, `call regex(pattern, batch, from, to)` -- copy
!     the results back, while `call fx_batch_search(pattern, batch)` / `fx_batch_match` leave them on the device for
!     fx_batch_count (a reduction on the device) and fx_batch_fetch.  The host-buffer specifics above move every row over PCIe on
!     every call (about 45 GB/s); a resident batch is scanned at the kernels' rate.
!   * There is no CPU matching path: without a HIP device the calls stop with an error message.
#if defined(IMPURE)
#define PURE_
#define ELEMENTAL_ impure elemental
#else
#define PURE_ pure
#define ELEMENTAL_ pure elemental
#endif
module forgex
   use, intrinsic :: iso_c_binding
   implicit none
   private

   public :: is_valid_regex
   public :: operator(.in.)
   public :: operator(.match.)
   public :: regex
   public :: regex_f
   ! device-resident batches (an extension of the reference's surface, src/forgex.F90:24-54: the same generics accept a batch)
   public :: fx_batch
   public :: fx_batch_upload, fx_batch_wrap, fx_batch_free, fx_batch_size
   public :: fx_batch_search, fx_batch_match, fx_batch_sync, fx_batch_count, fx_batch_fetch

   !> rows resident in HBM (uploaded once, or a caller's device pointer); results of the last run stay on the device with it
   type :: fx_batch
      private
      type(c_ptr) :: h = c_null_ptr
      integer :: n = 0
      integer :: row_len = 0
   end type fx_batch

   integer(c_int), parameter :: FXAMD_OP_SEARCH = 0, FXAMD_OP_MATCH = 1
   integer, parameter :: INVALID_CHAR_INDEX = -9999

   ! The library's subroutine forms (include/forgex_amd.h, "Subroutine forms ... for Fortran pure hosts"): a PURE FUNCTION may only
   ! have INTENT(IN) / VALUE dummies (F2018 C1590 -- gfortran rejects anything else), and a pure function call whose result is not
   ! needed may be dropped, so every output, the return code included, is an INTENT(OUT) argument of a pure SUBROUTINE.
   interface
      PURE_ subroutine fxamd_f_compile(pattern, pattern_len, op, prog, status, rc) bind(C, name='fxamd_f_compile')
         import :: c_char, c_int64_t, c_int, c_ptr, c_int32_t
         character(kind=c_char), intent(in) :: pattern(*)
         integer(c_int64_t), value :: pattern_len
         integer(c_int), value :: op
         type(c_ptr), intent(out) :: prog
         integer(c_int32_t), intent(out) :: status
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      PURE_ subroutine fxamd_f_program_free(prog, rc) bind(C, name='fxamd_f_program_free')
         import :: c_ptr, c_int32_t
         type(c_ptr), value :: prog
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      PURE_ subroutine fxamd_f_match_batch_host(prog, rows, n, row_len, flags, from, to, rc) bind(C, name='fxamd_f_match_batch_host')
         import :: c_ptr, c_int64_t, c_int32_t
         type(c_ptr), value :: prog, rows, flags, from, to
         integer(c_int64_t), value :: n, row_len
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      subroutine fxamd_f_batch_upload(rows, n, row_len, batch, rc) bind(C, name='fxamd_f_batch_upload')
         import :: c_ptr, c_int64_t, c_int32_t
         type(c_ptr), value :: rows
         integer(c_int64_t), value :: n, row_len
         type(c_ptr), intent(out) :: batch
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      subroutine fxamd_f_batch_wrap(d_rows, n, row_len, batch, rc) bind(C, name='fxamd_f_batch_wrap')
         import :: c_ptr, c_int64_t, c_int32_t
         type(c_ptr), value :: d_rows
         integer(c_int64_t), value :: n, row_len
         type(c_ptr), intent(out) :: batch
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      subroutine fxamd_f_batch_free(batch, rc) bind(C, name='fxamd_f_batch_free')
         import :: c_ptr, c_int32_t
         type(c_ptr), value :: batch
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      subroutine fxamd_f_batch_run(progs, m, batch, with_spans, rc) bind(C, name='fxamd_f_batch_run')
         import :: c_ptr, c_int32_t
         type(c_ptr), intent(in) :: progs(*)
         integer(c_int32_t), value :: m
         type(c_ptr), value :: batch
         integer(c_int32_t), value :: with_spans
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      subroutine fxamd_f_batch_sync(batch, rc) bind(C, name='fxamd_f_batch_sync')
         import :: c_ptr, c_int32_t
         type(c_ptr), value :: batch
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      subroutine fxamd_f_batch_fetch(batch, which, flags, from, to, rc) bind(C, name='fxamd_f_batch_fetch')
         import :: c_ptr, c_int32_t
         type(c_ptr), value :: batch
         integer(c_int32_t), value :: which
         type(c_ptr), value :: flags, from, to
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      subroutine fxamd_f_batch_count(batch, which, n_matches, rc) bind(C, name='fxamd_f_batch_count')
         import :: c_ptr, c_int32_t, c_int64_t
         type(c_ptr), value :: batch
         integer(c_int32_t), value :: which
         integer(c_int64_t), intent(out) :: n_matches
         integer(c_int32_t), intent(out) :: rc
      end subroutine
      PURE_ subroutine fxamd_f_strerror_copy(status, buf, cap, n) bind(C, name='fxamd_f_strerror_copy')
         import :: c_int32_t, c_char, c_int64_t
         integer(c_int32_t), value :: status
         character(kind=c_char), intent(inout) :: buf(*)
         integer(c_int64_t), value :: cap
         integer(c_int64_t), intent(out) :: n
      end subroutine
   end interface

   interface is_valid_regex
      module procedure :: is_valid_regex_pattern
   end interface

   interface operator(.in.)
      module procedure :: operator__in
      module procedure :: operator__in_batch
      module procedure :: operator__in_patterns
      module procedure :: operator__in_resident
      module procedure :: operator__in_patterns_resident
   end interface

   interface operator(.match.)
      module procedure :: operator__match
      module procedure :: operator__match_batch
      module procedure :: operator__match_patterns
      module procedure :: operator__match_resident
      module procedure :: operator__match_patterns_resident
   end interface

   interface regex
      module procedure :: subroutine__regex
      module procedure :: subroutine__regex_batch
      module procedure :: subroutine__regex_resident
   end interface

   interface regex_f
      module procedure :: function__regex
   end interface regex_f

contains

   !> get_error_message of the reference (src/essential/error_m.F90:127-211), from the library's table
   PURE_ function error_message(code) result(msg)
      integer, intent(in) :: code
      character(:), allocatable :: msg
      character(kind=c_char) :: buf(256)
      integer(c_int64_t) :: n
      integer :: i
      call fxamd_f_strerror_copy(int(code, c_int32_t), buf, int(size(buf), c_int64_t), n)
      allocate(character(int(n)) :: msg)
      do i = 1, int(n)
         msg(i:i) = buf(i)
      end do
   end function error_message

   PURE_ subroutine compile(pattern, op, prog, status)
      character(*), intent(in) :: pattern
      integer(c_int), intent(in) :: op
      type(c_ptr), intent(out) :: prog
      integer, intent(out) :: status
      integer(c_int32_t) :: st
      integer(c_int32_t) :: rc
      character(kind=c_char) :: buf(max(1, len(pattern)))
      integer :: i
      do i = 1, len(pattern)
         buf(i) = pattern(i:i)
      end do
      call fxamd_f_compile(buf, int(len(pattern), c_int64_t), op, prog, st, rc)
      if (rc /= 0) error stop 'forgex (amd): fxamd_compile failed'
      status = int(st)
      ! where the reference itself would `error stop` (tree / NFA / DFA limits, SURVEY.md section 5)
      if (status >= 100) error stop 'forgex (amd): pattern exceeds the limits of the automaton builder (status >= 100)'
   end subroutine compile

   !> drop the handle's reference (the return code is an output, so the call cannot be elided)
   PURE_ subroutine release(prog)
      type(c_ptr), intent(in) :: prog
      integer(c_int32_t) :: rc
      call fxamd_f_program_free(prog, rc)
      if (rc /= 0) error stop 'forgex (amd): fxamd_program_free failed'
   end subroutine release

   !> run one batch: rows = storage of character(row_len) :: s(n)
   PURE_ subroutine run_batch(prog, rows, n, row_len, flags, from, to)
      type(c_ptr), intent(in) :: prog, rows
      integer, intent(in) :: n, row_len
      integer(c_int8_t), intent(inout), target :: flags(:)
      integer(c_int32_t), intent(inout), target, optional :: from(:), to(:)
      integer(c_int32_t) :: rc
      if (n == 0) return
      if (present(from) .and. present(to)) then
         call fxamd_f_match_batch_host(prog, rows, int(n, c_int64_t), int(row_len, c_int64_t), c_loc(flags), c_loc(from), c_loc(to), rc)
      else
         call fxamd_f_match_batch_host(prog, rows, int(n, c_int64_t), int(row_len, c_int64_t), c_loc(flags), c_null_ptr, c_null_ptr, rc)
      end if
      if (rc /= 0) error stop 'forgex (amd): fxamd_match_batch_host failed (the match path needs a HIP device; there is no CPU fallback)'
   end subroutine run_batch

   ELEMENTAL_ function is_valid_regex_pattern(pattern) result(res)
      character(*), intent(in) :: pattern
      logical :: res
      type(c_ptr) :: prog
      integer(c_int32_t) :: st
      integer(c_int32_t) :: rc
      character(kind=c_char) :: buf(max(1, len(pattern)))
      integer :: i
      do i = 1, len(pattern)
         buf(i) = pattern(i:i)
      end do
      call fxamd_f_compile(buf, int(len(pattern), c_int64_t), FXAMD_OP_SEARCH, prog, st, rc)
      if (rc /= 0) error stop 'forgex (amd): fxamd_compile failed'
      res = st == 0 .or. st >= 100     ! (beyond the builder's limits is still a VALID pattern)
      call release(prog)
   end function is_valid_regex_pattern

   !---------------------------------------------------------------------------------------------------------------
   ! batch specifics: pattern compiled once, all rows in one call
   PURE_ function operator__in_batch(pattern, str) result(res)
      character(*), intent(in) :: pattern
      character(*), intent(in), target, contiguous :: str(:)
      logical :: res(size(str))
      res = flags_batch(pattern, str, FXAMD_OP_SEARCH)
   end function operator__in_batch

   PURE_ function operator__match_batch(pattern, str) result(res)
      character(*), intent(in) :: pattern
      character(*), intent(in), target, contiguous :: str(:)
      logical :: res(size(str))
      res = flags_batch(pattern, str, FXAMD_OP_MATCH)
   end function operator__match_batch

   PURE_ function flags_batch(pattern, str, op) result(res)
      character(*), intent(in) :: pattern
      character(*), intent(in), target, contiguous :: str(:)
      integer(c_int), intent(in) :: op
      logical :: res(size(str))
      type(c_ptr) :: prog
      integer :: status
      integer(c_int8_t), allocatable, target :: flags(:)
      res = .false.
      call compile(pattern, op, prog, status)
      if (status == 0 .and. size(str) > 0) then   ! invalid pattern => .false. everywhere (reference forgex.F90:101-104)
         allocate(flags(size(str)))
         flags = 0
         call run_batch(prog, c_loc(str), size(str), len(str), flags)
         res = flags /= 0
      end if
      call release(prog)
   end function flags_batch

   !---------------------------------------------------------------------------------------------------------------
   ! an ARRAY of patterns against an array of rows, element by element (the elemental contract of the reference's operators,
   ! src/forgex.F90:74,163, on two conformable arrays): every DISTINCT pattern is compiled once and meets its rows in one batch
   PURE_ function operator__in_patterns(pattern, str) result(res)
      character(*), intent(in) :: pattern(:)
      character(*), intent(in) :: str(:)
      logical :: res(size(str))
      res = flags_patterns(pattern, str, FXAMD_OP_SEARCH)
   end function operator__in_patterns

   PURE_ function operator__match_patterns(pattern, str) result(res)
      character(*), intent(in) :: pattern(:)
      character(*), intent(in) :: str(:)
      logical :: res(size(str))
      res = flags_patterns(pattern, str, FXAMD_OP_MATCH)
   end function operator__match_patterns

   PURE_ function flags_patterns(pattern, str, op) result(res)
      character(*), intent(in) :: pattern(:)
      character(*), intent(in) :: str(:)
      integer(c_int), intent(in) :: op
      logical :: res(size(str))
      logical :: done(size(str))
      character(len(str)), allocatable, target :: rows(:)
      integer, allocatable :: idx(:)
      logical, allocatable :: r(:)
      integer :: i, j, m
      if (size(pattern) /= size(str)) error stop 'forgex (amd): pattern and str arrays do not conform'
      done = .false.
      res = .false.
      do i = 1, size(str)
         if (done(i)) cycle
         ! rows that share pattern(i) (compared as the library sees them: byte for byte, trailing blanks included)
         m = 0
         allocate(idx(size(str) - i + 1))
         do j = i, size(str)
            if (.not. done(j)) then
               if (pattern(j) == pattern(i)) then
                  m = m + 1
                  idx(m) = j
                  done(j) = .true.
               end if
            end if
         end do
         allocate(rows(m))
         do j = 1, m
            rows(j) = str(idx(j))
         end do
         r = flags_batch(pattern(i), rows, op)
         do j = 1, m
            res(idx(j)) = r(j)
         end do
         deallocate(rows, idx, r)
      end do
   end function flags_patterns

   !---------------------------------------------------------------------------------------------------------------
   ! elemental specifics (scalars and arrays of any other shape): one row per call
   ELEMENTAL_ function operator__in(pattern, str) result(res)
      character(*), intent(in) :: pattern, str
      logical :: res
      res = flag_one(pattern, str, FXAMD_OP_SEARCH)
   end function operator__in

   ELEMENTAL_ function operator__match(pattern, str) result(res)
      character(*), intent(in) :: pattern, str
      logical :: res
      res = flag_one(pattern, str, FXAMD_OP_MATCH)
   end function operator__match

   PURE_ function flag_one(pattern, str, op) result(res)
      character(*), intent(in) :: pattern, str
      integer(c_int), intent(in) :: op
      logical :: res
      character(len(str)), target :: row(1)
      logical :: r(1)
      row(1) = str
      r = flags_batch(pattern, row, op)
      res = r(1)
   end function flag_one

   !---------------------------------------------------------------------------------------------------------------
   !> `call regex(pattern, text, res, length, from, to, status, err_msg)` -- reference src/forgex.F90:235-347
   PURE_ subroutine subroutine__regex(pattern, text, res, length, from, to, status, err_msg)
      character(*),              intent(in)    :: pattern, text
      character(:), allocatable, intent(inout) :: res
      integer, optional,         intent(inout) :: length, from, to, status
      character(*), optional,    intent(inout) :: err_msg
      character(len(text)), target :: row(1)
      integer :: f(1), t(1), st
      row(1) = text
      call subroutine__regex_batch(pattern, row, f, t, st)
      if (present(status)) status = st
      if (present(err_msg)) err_msg = error_message(st)
      if (st /= 0) then           ! reference forgex.F90:266-274
         res = ''
         if (present(length)) length = 0
         if (present(from)) from = INVALID_CHAR_INDEX
         if (present(to)) to = INVALID_CHAR_INDEX
         return
      end if
      if (f(1) > 0 .and. t(1) > 0) then
         res = text(f(1):t(1))
         if (present(length)) length = t(1) - f(1) + 1
         if (present(from)) from = f(1)
         if (present(to)) to = t(1)
      else
         res = ''
         if (present(length)) length = 0
         if (present(from)) from = 0
         if (present(to)) to = 0
      end if
   end subroutine subroutine__regex

   !> batch form: 1-based byte spans of the leftmost-longest match of every row (0,0 = none; -9999 = invalid pattern)
   PURE_ subroutine subroutine__regex_batch(pattern, text, from, to, status)
      character(*), intent(in) :: pattern
      character(*), intent(in), target, contiguous :: text(:)
      integer, intent(inout) :: from(:), to(:)
      integer, optional, intent(inout) :: status
      type(c_ptr) :: prog
      integer :: st
      integer(c_int8_t), allocatable, target :: flags(:)
      integer(c_int32_t), allocatable, target :: f(:), t(:)
      ! (the pattern goes to the library untrimmed: fxamd_compile applies the reference's own TRIM / ^ / $ handling)
      call compile(pattern, FXAMD_OP_SEARCH, prog, st)
      if (present(status)) status = st
      if (st /= 0) then
         from = INVALID_CHAR_INDEX
         to = INVALID_CHAR_INDEX
      else if (size(text) > 0) then
         allocate(flags(size(text)), f(size(text)), t(size(text)))
         flags = 0; f = 0; t = 0
         call run_batch(prog, c_loc(text), size(text), len(text), flags, f, t)
         from = int(f)
         to = int(t)
      end if
      call release(prog)
   end subroutine subroutine__regex_batch

   !---------------------------------------------------------------------------------------------------------------
   ! device-resident batches
   !> upload the storage of `character(L) :: str(n)` once; free with fx_batch_free
   function fx_batch_upload(str) result(batch)
      character(*), intent(in), target, contiguous :: str(:)
      type(fx_batch) :: batch
      integer(c_int32_t) :: rc
      if (size(str) > 0) then
         call fxamd_f_batch_upload(c_loc(str), int(size(str), c_int64_t), int(len(str), c_int64_t), batch%h, rc)
      else   ! (c_loc of a zero-sized array is not defined)
         call fxamd_f_batch_upload(c_null_ptr, 0_c_int64_t, int(len(str), c_int64_t), batch%h, rc)
      end if
      if (rc /= 0) error stop 'forgex (amd): fxamd_batch_upload failed (the match path needs a HIP device; there is no CPU fallback)'
      batch%n = size(str)
      batch%row_len = len(str)
   end function fx_batch_upload

   !> wrap n rows of row_len bytes that already live in device memory (the caller keeps owning them)
   function fx_batch_wrap(d_rows, n, row_len) result(batch)
      type(c_ptr), intent(in) :: d_rows
      integer, intent(in) :: n, row_len
      type(fx_batch) :: batch
      integer(c_int32_t) :: rc
      call fxamd_f_batch_wrap(d_rows, int(n, c_int64_t), int(row_len, c_int64_t), batch%h, rc)
      if (rc /= 0) error stop 'forgex (amd): fxamd_batch_wrap failed'
      batch%n = n
      batch%row_len = row_len
   end function fx_batch_wrap

   subroutine fx_batch_free(batch)
      type(fx_batch), intent(inout) :: batch
      integer(c_int32_t) :: rc
      if (c_associated(batch%h)) call fxamd_f_batch_free(batch%h, rc)
      batch%h = c_null_ptr
      batch%n = 0
      batch%row_len = 0
   end subroutine fx_batch_free

   pure function fx_batch_size(batch) result(n)
      type(fx_batch), intent(in) :: batch
      integer :: n
      n = batch%n
   end function fx_batch_size

   !> m patterns over the resident rows; the results stay on the device.  status(i) /= 0: pattern i is invalid -- its rows read
   !> "no match" (reference forgex.F90:101-104); ok = .false. when that is so for at least one pattern
   subroutine run_resident(pattern, batch, op, spans, status)
      character(*), intent(in) :: pattern(:)
      type(fx_batch), intent(in) :: batch
      integer(c_int), intent(in) :: op
      logical, intent(in) :: spans
      integer, intent(out) :: status(:)
      type(c_ptr) :: progs(size(pattern))
      integer(c_int32_t) :: rc
      integer :: i
      if (.not. c_associated(batch%h)) error stop 'forgex (amd): the batch is not resident (fx_batch_upload / fx_batch_wrap first)'
      do i = 1, size(pattern)
         call compile(pattern(i), op, progs(i), status(i))
      end do
      ! (an invalid pattern's program fills its result set with "no match": the library handles it like any other)
      call fxamd_f_batch_run(progs, int(size(pattern), c_int32_t), batch%h, merge(1_c_int32_t, 0_c_int32_t, spans), rc)
      if (rc /= 0) error stop 'forgex (amd): fxamd_batch_run failed'
      do i = 1, size(pattern)
         call release(progs(i))
      end do
   end subroutine run_resident

   !> `.in.` / regex over the resident rows, results LEFT ON THE DEVICE (fx_batch_count, fx_batch_fetch); asynchronous
   subroutine fx_batch_search(pattern, batch, spans, status)
      character(*), intent(in) :: pattern
      type(fx_batch), intent(in) :: batch
      logical, intent(in), optional :: spans
      integer, intent(out), optional :: status
      integer :: st(1)
      logical :: sp
      sp = .true.
      if (present(spans)) sp = spans
      call run_resident([character(len(pattern)) :: pattern], batch, FXAMD_OP_SEARCH, sp, st)
      if (present(status)) status = st(1)
   end subroutine fx_batch_search

   !> `.match.` over the resident rows, verdicts left on the device
   subroutine fx_batch_match(pattern, batch, status)
      character(*), intent(in) :: pattern
      type(fx_batch), intent(in) :: batch
      integer, intent(out), optional :: status
      integer :: st(1)
      call run_resident([character(len(pattern)) :: pattern], batch, FXAMD_OP_MATCH, .false., st)
      if (present(status)) status = st(1)
   end subroutine fx_batch_match

   subroutine fx_batch_sync(batch)
      type(fx_batch), intent(in) :: batch
      integer(c_int32_t) :: rc
      call fxamd_f_batch_sync(batch%h, rc)
      if (rc /= 0) error stop 'forgex (amd): fxamd_batch_sync failed'
   end subroutine fx_batch_sync

   !> matching rows of result set `which` (1-based, default 1) of the last run: a reduction on the device
   function fx_batch_count(batch, which) result(n)
      type(fx_batch), intent(in) :: batch
      integer, intent(in), optional :: which
      integer(c_int64_t) :: n
      integer(c_int32_t) :: rc
      integer :: w
      w = 1
      if (present(which)) w = which
      call fxamd_f_batch_count(batch%h, int(w - 1, c_int32_t), n, rc)
      if (rc /= 0) error stop 'forgex (amd): fxamd_batch_count failed'
   end function fx_batch_count

   !> result set `which` of the last run copied to host arrays (flags: 0 / 1; from / to: 1-based byte spans, 0 = none)
   subroutine fx_batch_fetch(batch, flags, from, to, which)
      type(fx_batch), intent(in) :: batch
      integer(c_int8_t), intent(out), target, contiguous :: flags(:)
      integer(c_int32_t), intent(out), target, contiguous, optional :: from(:), to(:)
      integer, intent(in), optional :: which
      integer(c_int32_t) :: rc
      integer :: w
      w = 1
      if (present(which)) w = which
      if (size(flags) < batch%n) error stop 'forgex (amd): fx_batch_fetch: flags(:) is shorter than the batch'
      if (present(from) .and. present(to)) then
         if (size(from) < batch%n .or. size(to) < batch%n) error stop 'forgex (amd): fx_batch_fetch: from(:) / to(:) are shorter than the batch'
         call fxamd_f_batch_fetch(batch%h, int(w - 1, c_int32_t), c_loc(flags), c_loc(from), c_loc(to), rc)
      else
         call fxamd_f_batch_fetch(batch%h, int(w - 1, c_int32_t), c_loc(flags), c_null_ptr, c_null_ptr, rc)
      end if
      if (rc /= 0) error stop 'forgex (amd): fxamd_batch_fetch failed'
   end subroutine fx_batch_fetch

   !> the operators with a resident batch in place of the character array: verdicts copied back
   function operator__in_resident(pattern, batch) result(res)
      character(*), intent(in) :: pattern
      type(fx_batch), intent(in) :: batch
      logical :: res(batch%n)
      res = flags_resident(pattern, batch, FXAMD_OP_SEARCH)
   end function operator__in_resident

   function operator__match_resident(pattern, batch) result(res)
      character(*), intent(in) :: pattern
      type(fx_batch), intent(in) :: batch
      logical :: res(batch%n)
      res = flags_resident(pattern, batch, FXAMD_OP_MATCH)
   end function operator__match_resident

   function flags_resident(pattern, batch, op) result(res)
      character(*), intent(in) :: pattern
      type(fx_batch), intent(in) :: batch
      integer(c_int), intent(in) :: op
      logical :: res(batch%n)
      integer(c_int8_t), allocatable, target :: flags(:)
      integer :: st(1)
      call run_resident([character(len(pattern)) :: pattern], batch, op, .false., st)
      allocate(flags(max(1, batch%n)))
      call fx_batch_fetch(batch, flags)
      res = flags(1:batch%n) /= 0
   end function flags_resident

   !> an array of patterns against ONE resident batch: res(i, j) = pattern(j) .in. row i (every row meets every pattern: the rows
   !> are read from HBM once for the patterns that share a pass)
   function operator__in_patterns_resident(pattern, batch) result(res)
      character(*), intent(in) :: pattern(:)
      type(fx_batch), intent(in) :: batch
      logical :: res(batch%n, size(pattern))
      res = flags_patterns_resident(pattern, batch, FXAMD_OP_SEARCH)
   end function operator__in_patterns_resident

   function operator__match_patterns_resident(pattern, batch) result(res)
      character(*), intent(in) :: pattern(:)
      type(fx_batch), intent(in) :: batch
      logical :: res(batch%n, size(pattern))
      res = flags_patterns_resident(pattern, batch, FXAMD_OP_MATCH)
   end function operator__match_patterns_resident

   function flags_patterns_resident(pattern, batch, op) result(res)
      character(*), intent(in) :: pattern(:)
      type(fx_batch), intent(in) :: batch
      integer(c_int), intent(in) :: op
      logical :: res(batch%n, size(pattern))
      integer(c_int8_t), allocatable, target :: flags(:)
      integer :: st(size(pattern)), j
      if (size(pattern) == 0) return
      call run_resident(pattern, batch, op, .false., st)
      allocate(flags(max(1, batch%n)))
      do j = 1, size(pattern)
         call fx_batch_fetch(batch, flags, which=j)
         res(:, j) = flags(1:batch%n) /= 0
      end do
   end function flags_patterns_resident

   !> `call regex(pattern, batch, from, to [, status])`: 1-based byte spans of every resident row (0,0 = none; -9999 = invalid pattern)
   subroutine subroutine__regex_resident(pattern, batch, from, to, status)
      character(*), intent(in) :: pattern
      type(fx_batch), intent(in) :: batch
      integer, intent(inout) :: from(:), to(:)
      integer, optional, intent(inout) :: status
      integer(c_int8_t), allocatable, target :: flags(:)
      integer(c_int32_t), allocatable, target :: f(:), t(:)
      integer :: st(1)
      call run_resident([character(len(pattern)) :: pattern], batch, FXAMD_OP_SEARCH, .true., st)
      if (present(status)) status = st(1)
      if (st(1) /= 0) then
         from = INVALID_CHAR_INDEX
         to = INVALID_CHAR_INDEX
         return
      end if
      allocate(flags(max(1, batch%n)), f(max(1, batch%n)), t(max(1, batch%n)))
      call fx_batch_fetch(batch, flags, f, t)
      from(1:batch%n) = int(f(1:batch%n))
      to(1:batch%n) = int(t(1:batch%n))
   end subroutine subroutine__regex_resident

   PURE_ function function__regex(pattern, text) result(res)
      character(*), intent(in)  :: pattern, text
      character(:), allocatable :: res
      call subroutine__regex(pattern, text, res)
   end function function__regex

end module forgex
