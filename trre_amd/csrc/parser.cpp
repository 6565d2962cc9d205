// parser.cpp — trre pattern -> AST.
//
// Operator-precedence parse with an operator stack and an operand stack, one
// bit of state ("next token must be an operand" / "... an operator").  The
// grammar has no formal definition; its behaviour — including the way implicit
// epsilons are synthesised and when pending operators are reduced — is the one
// of the reference's parser (trre_nft.c:11-288, identical in trre_dft.c:13-288)
// because the accepted language *is* that behaviour.  Inputs on which the
// reference reads outside its buffers are rejected with kErrUndefined.
#include "front.hpp"

namespace trre {
namespace {

class PatternParser {
public:
    explicit PatternParser(const std::string& text) : src_(text) {}

    Ast run() {
        bool need_operand = true;
        size_t i = 0;
        const size_t n = src_.size();
        while (i < n) {
            const uint8_t c = at(i);
            if (need_operand) {
                if (c == '(') {
                    shift(c);
                } else if (c == '[') {
                    shift(c);
                    i = bracket_list(i + 1);
                    need_operand = false;
                } else if (c == '\\') {
                    if (i + 1 >= n)
                        throw Error(kErrUndefined, "error: trailing backslash (reads past the pattern in the reference)");
                    ++i;
                    operand(ast_.add('c', -1, -1, at(i)));
                    need_operand = false;
                } else if (c == '.') {
                    // any byte: the range 0..255 (trre_nft.c:215-221)
                    int32_t lo = ast_.add('c', -1, -1, 0), hi = ast_.add('c', -1, -1, 255);
                    operand(ast_.add('-', lo, hi));
                    need_operand = false;
                } else if (c == ':') {
                    // epsilon as the implicit left operand; the ':' is re-read as an operator
                    operand(ast_.add('e', -1, -1, c));
                    need_operand = false;
                    continue;
                } else if (c == '|' || c == '*' || c == '+' || c == '?' || c == ')' || c == '{' || c == '}') {
                    if (!ops_.empty() && ops_.back() == ':') {
                        // epsilon as the implicit right operand of a pending ':'
                        operand(ast_.add('e', -1, -1, c));
                        need_operand = false;
                        continue;
                    }
                    throw Error(kErrSyntax, std::string("error: unexpected symbol ") + (char)c);
                } else {
                    operand(ast_.add('c', -1, -1, c));
                    need_operand = false;
                }
            } else {
                if (c == '*' || c == '+' || c == '?') {
                    const bool lazy = (i + 1 < n && at(i + 1) == '?');
                    postfix(c, lazy);
                    if (lazy) ++i;
                } else if (c == '|') {
                    infix(c);
                    need_operand = true;
                } else if (c == ':') {
                    // a ':' that closes the pattern gets its epsilon BEFORE pending
                    // operators are reduced (trre_nft.c:254-260) — this ordering is
                    // observable: "ab:" means a -> "b:" .
                    if (i + 1 >= n) operand(ast_.add('e', -1, -1, c));
                    infix(c);
                    need_operand = true;
                } else if (c == '{') {
                    i = iteration(i + 1);
                } else if (c == ')') {
                    while (!ops_.empty() && ops_.back() != '(') fold();
                    if (ops_.empty()) throw Error(kErrSyntax, "error: unmached parenthesis");
                    ops_.pop_back();
                } else {
                    infix('.');           // juxtaposition
                    need_operand = true;
                    continue;             // the same byte is now read as an operand
                }
            }
            ++i;
        }
        while (!ops_.empty()) fold();
        if (vals_.empty())
            throw Error(kErrUndefined, "error: empty expression (assertion failure in the reference)");
        ast_.root = vals_.back();
        return std::move(ast_);
    }

private:
    static constexpr size_t kStackDepth = 1024;   // trre_nft.c:38-39 (unchecked there)

    static int binding(uint8_t op) {              // trre_nft.c:11-22
        switch (op) {
        case '|': return 1;
        case '-': return 2;
        case ':': return 3;
        case '.': return 4;
        case '?': case '*': case '+': case 'I': return 5;
        case '\\': return 6;
        default: return -1;
        }
    }

    uint8_t at(size_t i) const { return (uint8_t)src_[i]; }

    void shift(uint8_t op) {
        if (ops_.size() >= kStackDepth)
            throw Error(kErrUndefined, "error: operator stack overflow (undefined in the reference)");
        ops_.push_back(op);
    }
    void operand(int32_t node) {
        if (vals_.size() >= kStackDepth)
            throw Error(kErrUndefined, "error: operand stack overflow (undefined in the reference)");
        vals_.push_back(node);
    }
    int32_t take() {
        if (vals_.empty())
            throw Error(kErrUndefined, "error: operand stack underflow (undefined in the reference)");
        int32_t v = vals_.back();
        vals_.pop_back();
        return v;
    }

    // apply the operator on top of the stack (trre_nft.c:93-108)
    void fold() {
        const uint8_t op = ops_.back();
        ops_.pop_back();
        if (op == '|' || op == '.' || op == ':' || op == '-') {
            int32_t rhs = take();
            int32_t lhs = take();
            operand(ast_.add(op, lhs, rhs));
        } else if (op == '(') {
            throw Error(kErrSyntax, "error: unmached parenthesis");
        }
    }
    // left-associative binary operator (trre_nft.c:111-115)
    void infix(uint8_t op) {
        while (!ops_.empty() && binding(ops_.back()) >= binding(op)) fold();
        shift(op);
    }
    // * + ? bind to the operand just read (trre_nft.c:76-90)
    void postfix(uint8_t op, bool lazy) {
        int32_t body = take();
        operand(ast_.add(op, body, -1, lazy ? 1 : 0));
    }

    // "{m}", "{m,}", "{,n}", "{m,n}" with an optional trailing '?'
    // (trre_nft.c:117-156).  Returns the index of the last byte consumed.
    size_t iteration(size_t i) {
        int commas = 0, number = 0, lower = 0;
        for (; i < src_.size(); ++i) {
            const uint8_t c = at(i);
            if (c >= '0' && c <= '9') {
                number = number * 10 + (c - '0');
            } else if (c == ',') {
                lower = number;
                number = 0;
                ++commas;
            } else if (c == '}') {
                bool lazy = false;
                if (i + 1 < src_.size() && at(i + 1) == '?') { lazy = true; ++i; }
                if (commas == 0) lower = number;
                else if (commas > 1)
                    throw Error(kErrSyntax, "error: more then one comma in curly brackets");
                int32_t bounds = ast_.add((uint8_t)lower, -1, -1, (uint8_t)number);
                int32_t body = take();
                operand(ast_.add('I', body, bounds, lazy ? 1 : 0));
                return i;
            } else {
                throw Error(kErrSyntax, std::string("error: unexpected symbol in curly brackets: ") + (char)c);
            }
        }
        throw Error(kErrSyntax, "error: unmached curly brackets");
    }

    // "[...]": single-byte operands, explicit ':' and '-', juxtaposition means
    // alternation (trre_nft.c:158-193).  Returns the index of the ']'.
    size_t bracket_list(size_t i) {
        bool need_operand = true;
        while (i < src_.size()) {
            const uint8_t c = at(i);
            if (need_operand) {
                if (c == ':' || c == '-' || c == '[' || c == ']')
                    throw Error(kErrSyntax, std::string("error: unexpected symbol in square brackets: ") + (char)c);
                operand(ast_.add('c', -1, -1, c));
                need_operand = false;
            } else if (c == ':' || c == '-') {
                infix(c);
                need_operand = true;
            } else if (c == ']') {
                while (!ops_.empty() && ops_.back() != '[') fold();
                if (ops_.empty())
                    throw Error(kErrUndefined, "error: bracket marker lost (undefined in the reference)");
                ops_.pop_back();
                return i;
            } else {
                infix('|');
                need_operand = true;
                continue;
            }
            ++i;
        }
        throw Error(kErrSyntax, "error: unmached square brackets");
    }

    const std::string& src_;
    Ast ast_;
    std::vector<uint8_t> ops_;
    std::vector<int32_t> vals_;
};

}  // namespace

Ast parse_pattern(const std::string& pattern) {
    if (pattern.find('\0') != std::string::npos)
        throw Error(kErrArg, "error: pattern contains a NUL byte");
    return PatternParser(pattern).run();
}

}  // namespace trre
