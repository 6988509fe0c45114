! Test infrastructure only -- NOT part of the product.
!
! A RECORDING stand-in for the reference's test-harness module: it exposes the same public
! names the reference's test programs `use` (runner_in, runner_match, runner_regex,
! runner_prefix, runner_suffix, runner_validate, runner_error, nchar, is_eqv_str, print_hex;
! interface of reference src/test_m.F90:19-38, :199-395) but, instead of asserting, it calls
! the REAL reference API and appends one record per call to $FORGEX_RECORD_FILE.  Compiling the
! reference's own test programs (where they lie under /root/reference/test) against this
! module therefore turns every one of their assertions into a golden vector holding the
! test's input, the answer the test expects, and what the reference actually returns.
!
! Record format (tab separated; byte strings hex encoded, '-' = empty string):
!   in       <pat> <text> <expected T|F> <got T|F> <from> <to> <length> <status>
!   match    <pat> <text> <expected T|F> <got T|F>
!   regex    <pat> <text> <expected_substr> <got_substr> <from> <to> <length> <status>
!   prefix   <pat> <expected> <got>
!   suffix   <pat> <expected> <got>
!   validate <pat> <expected T|F> <got T|F>
!   error    <pat> <expected_code> <got_code> <got_message>
module forgex_test_m
   use, intrinsic :: iso_fortran_env, only: int8, int32, error_unit, output_unit
   use :: forgex, only: operator(.in.), operator(.match.), regex, is_valid_regex
   use :: forgex_syntax_tree_graph_m, only: tree_t
   implicit none
   private

   public :: runner_validate, runner_in, runner_match, runner_regex
   public :: runner_prefix, runner_suffix, runner_error
   public :: nchar, is_eqv_str, print_hex

   character(1), parameter :: TAB = achar(9)

contains

   function tohex(s) result(h)
      character(*), intent(in) :: s
      character(:), allocatable :: h
      integer :: i
      if (len(s) == 0) then
         h = '-'
         return
      end if
      allocate(character(2*len(s)) :: h)
      do i = 1, len(s)
         write(h(2*i-1:2*i), '(z2.2)') iachar(s(i:i))
      end do
   end function tohex

   character(1) function tf(l)
      logical, intent(in) :: l
      tf = merge('T', 'F', l)
   end function tf

   function itoa(i) result(s)
      integer, intent(in) :: i
      character(:), allocatable :: s
      character(16) :: buf
      write(buf, '(i0)') i
      s = trim(buf)
   end function itoa

   subroutine emit(line)
      character(*), intent(in) :: line
      character(4096) :: path
      integer :: u, stat, l
      call get_environment_variable('FORGEX_RECORD_FILE', path, length=l, status=stat)
      if (stat /= 0 .or. l == 0) then
         write(output_unit, '(a)') line
         return
      end if
      open(newunit=u, file=path(1:l), position='append', action='write', status='unknown')
      write(u, '(a)') line
      close(u)
   end subroutine emit

   subroutine runner_validate(pattern, answer, result)
      character(*), intent(in)    :: pattern
      logical,      intent(in)    :: answer
      logical,      intent(inout) :: result
      logical :: got
      got = is_valid_regex(pattern)
      call emit('validate'//TAB//tohex(pattern)//TAB//tf(answer)//TAB//tf(got))
      result = result .and. (got .eqv. answer)
   end subroutine runner_validate

   subroutine runner_in(pattern, str, answer, result)
      character(*), intent(in)    :: pattern, str
      logical,      intent(in)    :: answer
      logical,      intent(inout) :: result
      logical :: got
      character(:), allocatable :: sub
      integer :: from, to, length, status
      got = pattern .in. str
      from = -1; to = -1; length = -1; status = -1
      sub = ''
      call regex(pattern, str, sub, length=length, from=from, to=to, status=status)
      call emit('in'//TAB//tohex(pattern)//TAB//tohex(str)//TAB//tf(answer)//TAB//tf(got)//TAB// &
                itoa(from)//TAB//itoa(to)//TAB//itoa(length)//TAB//itoa(status))
      result = result .and. (got .eqv. answer)
   end subroutine runner_in

   subroutine runner_match(pattern, str, answer, result)
      character(*), intent(in)    :: pattern, str
      logical,      intent(in)    :: answer
      logical,      intent(inout) :: result
      logical :: got
      got = pattern .match. str
      call emit('match'//TAB//tohex(pattern)//TAB//tohex(str)//TAB//tf(answer)//TAB//tf(got))
      result = result .and. (got .eqv. answer)
   end subroutine runner_match

   subroutine runner_regex(pattern, str, answer, result)
      character(*), intent(in)    :: pattern, str
      character(*), intent(in)    :: answer
      logical,      intent(inout) :: result
      character(:), allocatable :: sub
      integer :: from, to, length, status
      from = -1; to = -1; length = -1; status = -1
      sub = ''
      call regex(pattern, str, sub, length=length, from=from, to=to, status=status)
      call emit('regex'//TAB//tohex(pattern)//TAB//tohex(str)//TAB//tohex(answer)//TAB//tohex(sub)//TAB// &
                itoa(from)//TAB//itoa(to)//TAB//itoa(length)//TAB//itoa(status))
      result = result .and. is_eqv_str(sub, answer)
   end subroutine runner_regex

   subroutine runner_prefix(pattern, prefix, result)
      use :: forgex_syntax_tree_optimize_m, only: extract_literal
      character(*), intent(in) :: pattern, prefix
      logical, intent(inout) :: result
      character(:), allocatable :: all, pre, suf, fac
      type(tree_t) :: tree
      call tree%build(pattern)
      all = ''; pre = ''; suf = ''; fac = ''
      call extract_literal(tree, all, pre, suf, fac)
      call emit('prefix'//TAB//tohex(pattern)//TAB//tohex(prefix)//TAB//tohex(pre))
      result = result .and. (prefix == pre)
   end subroutine runner_prefix

   subroutine runner_suffix(pattern, suffix, result)
      use :: forgex_syntax_tree_optimize_m, only: extract_literal
      character(*), intent(in) :: pattern, suffix
      logical, intent(inout) :: result
      character(:), allocatable :: all, pre, suf, fac
      type(tree_t) :: tree
      call tree%build(pattern)
      all = ''; pre = ''; suf = ''; fac = ''
      call extract_literal(tree, all, pre, suf, fac)
      call emit('suffix'//TAB//tohex(pattern)//TAB//tohex(suffix)//TAB//tohex(suf))
      result = result .and. (suffix == suf)
   end subroutine runner_suffix

   subroutine runner_error(pattern, text, code, result)
      character(*), intent(in) :: pattern, text
      integer, intent(in) :: code
      logical, intent(inout) :: result
      integer(int32) :: status
      character(256) :: err_msg
      character(:), allocatable :: sub
      sub = ''
      status = -1
      err_msg = ''
      call regex(pattern, "", sub, status=status, err_msg=err_msg)
      call emit('error'//TAB//tohex(pattern)//TAB//itoa(code)//TAB//itoa(status)//TAB//tohex(trim(err_msg)))
      result = result .and. (status == code)
   end subroutine runner_error

   pure function nchar(i) result(chara)
      integer(int8), intent(in) :: i
      character(1) :: chara
      if (i < 0) then
         chara = char(i+256)
      else
         chara = char(i)
      end if
   end function nchar

   subroutine print_hex(str)
      character(*), intent(in) :: str
      ! recording harness: nothing to print
   end subroutine print_hex

   logical function is_eqv_str(str, ret)
      character(*), intent(in) :: str, ret
      integer :: j
      is_eqv_str = len(str) == len(ret)
      if (.not. is_eqv_str) return
      do j = 1, len(str)
         is_eqv_str = is_eqv_str .and. (str(j:j) == ret(j:j))
      end do
   end function is_eqv_str

end module forgex_test_m
