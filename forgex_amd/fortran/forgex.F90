! forgex -- drop-in Fortran module for the batch match path, MI355X build.
!
! Same public names as the reference module (reference src/forgex.F90:24-54):
!     is_valid_regex, operator(.in.), operator(.match.), regex, regex_f
! with the same argument meaning and error behaviour, bound through iso_c_binding to the C ABI of
! include/forgex_amd.h (libforgex_amd.so: host table compiler + HIP kernels for gfx950).
!
! Differences a maintainer should know (INTEGRATION.md):
!   * the specifics are IMPURE (they call the GPU library); the reference's own `IMPURE` build switch
!     (reference src/forgex.F90:10-13) is the precedent.  `impure elemental` keeps every array shape working.
!   * rank-1 character arrays resolve to NON-elemental batch specifics (a non-elemental specific is preferred over an
!     elemental one), so `pattern .in. strs(:)` compiles the pattern ONCE and matches all rows in one kernel launch
!     instead of re-parsing the pattern per element (reference src/forgex.F90:98,139-140).
!   * `regex` gains a batch form returning from(:)/to(:) for a rank-1 text array.
!   * there is no CPU matching path: without a HIP device the calls stop with an error message.
module forgex
   use, intrinsic :: iso_c_binding
   use, intrinsic :: iso_fortran_env, only: error_unit
   implicit none
   private

   public :: is_valid_regex
   public :: operator(.in.)
   public :: operator(.match.)
   public :: regex
   public :: regex_f

   integer(c_int), parameter :: FXAMD_OP_SEARCH = 0, FXAMD_OP_MATCH = 1
   integer, parameter :: INVALID_CHAR_INDEX = -9999

   interface
      function fxamd_compile(pattern, pattern_len, op, prog, status) bind(C, name='fxamd_compile') result(rc)
         import :: c_char, c_int64_t, c_int, c_ptr, c_int32_t
         character(kind=c_char), intent(in) :: pattern(*)
         integer(c_int64_t), value :: pattern_len
         integer(c_int), value :: op
         type(c_ptr), intent(out) :: prog
         integer(c_int32_t), intent(out) :: status
         integer(c_int) :: rc
      end function
      subroutine fxamd_program_free(prog) bind(C, name='fxamd_program_free')
         import :: c_ptr
         type(c_ptr), value :: prog
      end subroutine
      function fxamd_match_batch_host(prog, rows, n, row_len, flags, from, to) bind(C, name='fxamd_match_batch_host') result(rc)
         import :: c_ptr, c_int64_t, c_int
         type(c_ptr), value :: prog, rows, flags, from, to
         integer(c_int64_t), value :: n, row_len
         integer(c_int) :: rc
      end function
      function fxamd_strerror(status) bind(C, name='fxamd_strerror') result(msg)
         import :: c_int32_t, c_ptr
         integer(c_int32_t), value :: status
         type(c_ptr) :: msg
      end function
   end interface

   interface is_valid_regex
      module procedure :: is_valid_regex_pattern
   end interface

   interface operator(.in.)
      module procedure :: operator__in
      module procedure :: operator__in_batch
   end interface

   interface operator(.match.)
      module procedure :: operator__match
      module procedure :: operator__match_batch
   end interface

   interface regex
      module procedure :: subroutine__regex
      module procedure :: subroutine__regex_batch
   end interface

   interface regex_f
      module procedure :: function__regex
   end interface regex_f

contains

   function error_message(code) result(msg)
      integer, intent(in) :: code
      character(:), allocatable :: msg
      type(c_ptr) :: p
      character(kind=c_char), pointer :: s(:)
      integer :: n
      p = fxamd_strerror(int(code, c_int32_t))
      call c_f_pointer(p, s, [1024])
      n = 0
      do while (n < 1024)
         if (s(n+1) == c_null_char) exit
         n = n + 1
      end do
      allocate(character(n) :: msg)
      msg = transfer(s(1:n), msg)
   end function error_message

   subroutine compile(pattern, op, prog, status)
      character(*), intent(in) :: pattern
      integer(c_int), intent(in) :: op
      type(c_ptr), intent(out) :: prog
      integer, intent(out) :: status
      integer(c_int32_t) :: st
      integer(c_int) :: rc
      character(kind=c_char), allocatable :: buf(:)
      allocate(buf(max(1, len(pattern))))
      if (len(pattern) > 0) buf = transfer(pattern, buf)
      rc = fxamd_compile(buf, int(len(pattern), c_int64_t), op, prog, st)
      if (rc /= 0) then
         write(error_unit, '(a,i0)') 'forgex (amd): fxamd_compile failed, rc=', rc
         error stop
      end if
      status = int(st)
      if (status >= 100) then   ! where the reference itself would `error stop` (state limits, SURVEY.md section 5)
         write(error_unit, '(a)') 'forgex (amd): '//error_message(status)
         error stop
      end if
   end subroutine compile

   !> run one batch: rows = storage of character(row_len) :: s(n)
   subroutine run_batch(prog, rows, n, row_len, flags, from, to)
      type(c_ptr), intent(in) :: prog, rows
      integer, intent(in) :: n, row_len
      integer(c_int8_t), intent(inout), target :: flags(:)
      integer(c_int32_t), intent(inout), target, optional :: from(:), to(:)
      integer(c_int) :: rc
      if (n == 0) return
      if (present(from) .and. present(to)) then
         rc = fxamd_match_batch_host(prog, rows, int(n, c_int64_t), int(row_len, c_int64_t), c_loc(flags), c_loc(from), c_loc(to))
      else
         rc = fxamd_match_batch_host(prog, rows, int(n, c_int64_t), int(row_len, c_int64_t), c_loc(flags), c_null_ptr, c_null_ptr)
      end if
      if (rc /= 0) then
         write(error_unit, '(a,i0,a)') 'forgex (amd): fxamd_match_batch_host failed, rc=', rc, &
            ' (the match path needs a HIP device; there is no CPU fallback)'
         error stop
      end if
   end subroutine run_batch

   impure elemental function is_valid_regex_pattern(pattern) result(res)
      character(*), intent(in) :: pattern
      logical :: res
      type(c_ptr) :: prog
      integer :: status
      call compile(pattern, FXAMD_OP_SEARCH, prog, status)
      res = status == 0
      call fxamd_program_free(prog)
   end function is_valid_regex_pattern

   !---------------------------------------------------------------------------------------------------------------
   ! batch specifics: pattern compiled once, all rows in one launch
   function operator__in_batch(pattern, str) result(res)
      character(*), intent(in) :: pattern
      character(*), intent(in), target, contiguous :: str(:)
      logical :: res(size(str))
      res = flags_batch(pattern, str, FXAMD_OP_SEARCH)
   end function operator__in_batch

   function operator__match_batch(pattern, str) result(res)
      character(*), intent(in) :: pattern
      character(*), intent(in), target, contiguous :: str(:)
      logical :: res(size(str))
      res = flags_batch(pattern, str, FXAMD_OP_MATCH)
   end function operator__match_batch

   function flags_batch(pattern, str, op) result(res)
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
      call fxamd_program_free(prog)
   end function flags_batch

   !---------------------------------------------------------------------------------------------------------------
   ! elemental specifics (scalars and arrays of any other rank): one row per call
   impure elemental function operator__in(pattern, str) result(res)
      character(*), intent(in) :: pattern, str
      logical :: res
      res = flag_one(pattern, str, FXAMD_OP_SEARCH)
   end function operator__in

   impure elemental function operator__match(pattern, str) result(res)
      character(*), intent(in) :: pattern, str
      logical :: res
      res = flag_one(pattern, str, FXAMD_OP_MATCH)
   end function operator__match

   function flag_one(pattern, str, op) result(res)
      character(*), intent(in) :: pattern, str
      integer(c_int), intent(in) :: op
      logical :: res
      character(:), allocatable, target :: row(:)
      logical :: r(1)
      allocate(character(len(str)) :: row(1))
      row(1) = str
      r = flags_batch(pattern, row, op)
      res = r(1)
   end function flag_one

   !---------------------------------------------------------------------------------------------------------------
   !> `call regex(pattern, text, res, length, from, to, status, err_msg)` -- reference src/forgex.F90:235-347
   subroutine subroutine__regex(pattern, text, res, length, from, to, status, err_msg)
      character(*),              intent(in)    :: pattern, text
      character(:), allocatable, intent(inout) :: res
      integer, optional,         intent(inout) :: length, from, to, status
      character(*), optional,    intent(inout) :: err_msg
      character(:), allocatable, target :: row(:)
      integer :: f(1), t(1), st
      allocate(character(len(text)) :: row(1))
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
   subroutine subroutine__regex_batch(pattern, text, from, to, status)
      character(*), intent(in) :: pattern
      character(*), intent(in), target, contiguous :: text(:)
      integer, intent(inout) :: from(:), to(:)
      integer, optional, intent(inout) :: status
      type(c_ptr) :: prog
      integer :: st
      integer(c_int8_t), allocatable, target :: flags(:)
      integer(c_int32_t), allocatable, target :: f(:), t(:)
      call compile(trim_keep(pattern), FXAMD_OP_SEARCH, prog, st)
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
      call fxamd_program_free(prog)
   end subroutine subroutine__regex_batch

   !> the pattern goes to the library untrimmed: fxamd_compile applies the reference's own TRIM / ^ / $ handling
   pure function trim_keep(pattern) result(p)
      character(*), intent(in) :: pattern
      character(:), allocatable :: p
      p = pattern
   end function trim_keep

   function function__regex(pattern, text) result(res)
      character(*), intent(in)  :: pattern, text
      character(:), allocatable :: res
      call subroutine__regex(pattern, text, res)
   end function function__regex

end module forgex
