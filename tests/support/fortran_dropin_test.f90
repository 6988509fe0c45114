! Drop-in check of the Fortran module (GPU required).  Written like the reference's own test programs
! (reference test/test_api/test_case_001.f90, test_case_003.f90:27-28): `use forgex`, operators, regex.
program fortran_dropin_test
   use :: forgex
   implicit none
   logical :: ok
   character(:), allocatable :: res
   integer :: length, from, to, status, i
   character(256) :: msg
   character(8) :: rows(4)
   character(16) :: texts(3)
   logical :: flags4(4), flags22(2, 2)
   character(12) :: pats(4)
   integer :: f3(3), t3(3)

   ok = .true.
   ! scalar operators
   ok = ok .and. ('\d{3}-\d{4}' .match. '100-1002')
   ok = ok .and. .not. ('\d{3}-\d{4}' .match. '1234567')
   ok = ok .and. ('[a-z]+\d+' .in. 'ab12  cd345')
   ok = ok .and. ('ab[cd]' .match. 'ab')                 ! reference quirk (SURVEY Appendix A.7)
   ok = ok .and. .not. ('aa[bc]' .in. 'aaab')           ! reference quirk (Appendix A.5)
   ok = ok .and. .not. ('a(' .in. 'a(')                  ! invalid pattern => .false.
   ok = ok .and. is_valid_regex('foo(bar|baz)') .and. .not. is_valid_regex('a{2,1}')
   if (.not. ok) print *, 'scalar operator checks FAILED'

   ! batch (rank-1) operators: one compile, one launch
   rows = [character(8) :: '100-1002', '1234567 ', '999-0000', 'abc-defg']
   flags4 = '\d{3}-\d{4}' .match. rows
   ok = ok .and. all(flags4 .eqv. [.true., .false., .true., .false.])
   flags4 = '\d' .in. rows
   ok = ok .and. all(flags4 .eqv. [.true., .true., .true., .false.])

   ! an ARRAY of patterns against an array of rows, element by element (the operators are elemental in the reference): every
   ! distinct pattern is compiled once (`.in.` trims the pattern: reference forgex.F90:95)
   pats = [character(12) :: '\d{3}-\d{4}', '[a-z]+', '\d{3}-\d{4}', 'abc']
   flags4 = pats .in. rows
   ok = ok .and. all(flags4 .eqv. [.true., .false., .true., .true.])
   ! rank-2 arrays still resolve to the elemental specifics
   flags22 = '\d' .in. reshape(rows, [2, 2])
   ok = ok .and. all(flags22 .eqv. reshape([.true., .true., .true., .false.], [2, 2]))

   ! regex subroutine, scalar
   call regex('foo(bar|baz)', 'xxfoobarbaz', res, length=length, from=from, to=to, status=status, err_msg=msg)
   ok = ok .and. res == 'foobar' .and. length == 6 .and. from == 3 .and. to == 8 .and. status == 0
   ok = ok .and. trim(msg) == 'Given pattern is valid.'
   call regex('a(', 'zz', res, length=length, from=from, to=to, status=status, err_msg=msg)
   ok = ok .and. res == '' .and. len(res) == 0 .and. length == 0 .and. from == -9999 .and. to == -9999 .and. status == 2
   ok = ok .and. trim(msg) == 'ERROR: Closing parenthesis is expected.'
   call regex('b*', 'aaa', res, length=length, from=from, to=to)
   ok = ok .and. len(res) == 0 .and. length == 0 .and. from == 0 .and. to == 0
   ok = ok .and. regex_f('[a-z]+\d+', 'ab12  cd345') == 'ab12'

   ! regex, batch form
   texts = [character(16) :: 'ab12  cd345     ', 'no digits here  ', '   z9           ']
   call regex('[a-z]+\d+', texts, f3, t3, status)
   ok = ok .and. all(f3 == [1, 0, 4]) .and. all(t3 == [4, 0, 5]) .and. status == 0

   if (ok) then
      print '(a)', 'FORTRAN DROP-IN OK'
   else
      print '(a)', 'FORTRAN DROP-IN FAILED'
      error stop 1
   end if
end program fortran_dropin_test
